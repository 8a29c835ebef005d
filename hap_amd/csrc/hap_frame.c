/*
 * hap_frame.c -- host-side Hap container logic (see hap_frame.h).
 *
 * Behavioural contract: the frame layout of documentation/HapVideoDRAFT.md
 * and the observable behaviour of /root/reference/source/hap.c (cited per
 * function), including its size arithmetic quirks, with one deliberate
 * hardening: chunk byte ranges are bounds-checked against the texture section
 * (the reference reads out of bounds, hap.c:798-809).
 */
#include "hap_frame.h"
#include "../../include/hap.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- bytes -- */
static uint32_t le24(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
static uint32_t le32(const uint8_t *p) { return le24(p) | ((uint32_t)p[3] << 24); }

void hapf_reader_init_host(hapf_reader *r, const void *frame, uint64_t length)
{
    memset(r, 0, sizeof(*r));
    r->view = (const uint8_t *)frame;
    r->view_len = length;
}

void hapf_reader_free(hapf_reader *r)
{
    free(r->side);
    r->side = NULL;
    r->side_cap = 0;
    r->side_len = 0;
}

const uint8_t *hapf_need(hapf_reader *r, uint64_t offset, uint64_t length)
{
    uint64_t want = length;
    if (offset + length <= r->view_len)
        return r->view + offset;
    if (r->view2 && offset >= r->view2_off && offset + length <= r->view2_off + r->view2_len)
        return r->view2 + (offset - r->view2_off);
    if (!r->fetch)
        return NULL;
    if (r->side && r->side_len && offset >= r->side_off && offset + length <= r->side_off + r->side_len)
        return r->side + (offset - r->side_off);
    /* a fetch costs a round trip to the device: bring the neighbourhood too (the tables of a section follow its header) */
    if (r->total_len && want < 4096u) {
        want = 4096u;
        if (offset + want > r->total_len)
            want = r->total_len > offset ? r->total_len - offset : 0u;
        if (want < length)
            want = length;
    }
    if (want > r->side_cap) {
        uint8_t *n = (uint8_t *)realloc(r->side, (size_t)want + 16);
        if (!n)
            return NULL;
        r->side = n;
        r->side_cap = want;
    }
    r->side_len = 0;
    if (r->fetch(r->user, offset, want, r->side) != 0)
        return NULL;
    r->side_off = offset;
    r->side_len = want;
    return r->side;
}

/* ------------------------------------------------------------- sections -- */

/* Section header: u24 length + type byte; a zero u24 means the length is the
 * u32 that follows (reference hap.c:137-187). `available` plays the role of
 * the reference's buffer_length, including its 32-bit wrap-around. */
int hapf_read_section(const uint8_t *p, uint32_t available, hapf_section *out)
{
    if (available < 4u)
        return HapResult_Bad_Frame;
    out->length = le24(p);
    out->header_len = 4u;
    if (out->length == 0u) {
        if (available < 8u)
            return HapResult_Bad_Frame;
        out->length = le32(p + 4);
        out->header_len = 8u;
    }
    out->type = p[3];
    /* hardening: compared in 64 bits.  The reference adds in 32 bits (hap.c:160-181), so a length of
       0xFFFFFFF8 and more wraps and is accepted there; such a section cannot exist in `available` bytes. */
    if ((uint64_t)out->header_len + out->length > available)
        return HapResult_Bad_Frame;
    return HapResult_No_Error;
}

static int read_section_at(hapf_reader *r, uint64_t offset, uint32_t available, hapf_section *out)
{
    const uint8_t *p;
    if (available < 4u)
        return HapResult_Bad_Frame;
    p = hapf_need(r, offset, available < 8u ? 4u : 8u);
    if (!p)
        return HapResult_Bad_Frame;
    return hapf_read_section(p, available, out);
}

/* reference hap.c:189-212 */
void hapf_write_section(uint8_t *p, unsigned header_len, uint32_t length, unsigned type)
{
    uint32_t first = header_len == 4u ? length : 0u;
    p[0] = (uint8_t)first;
    p[1] = (uint8_t)(first >> 8);
    p[2] = (uint8_t)(first >> 16);
    p[3] = (uint8_t)type;
    if (header_len != 4u) {
        p[4] = (uint8_t)length;
        p[5] = (uint8_t)(length >> 8);
        p[6] = (uint8_t)(length >> 16);
        p[7] = (uint8_t)(length >> 24);
    }
}

/* reference hap.c:215-261 */
static const struct { unsigned nibble, format; } k_formats[] = {
    {0xB, HapTextureFormat_RGB_DXT1},
    {0xE, HapTextureFormat_RGBA_DXT5},
    {0xF, HapTextureFormat_YCoCg_DXT5},
    {0x1, HapTextureFormat_A_RGTC1},
    {0xC, HapTextureFormat_RGBA_BPTC_UNORM},
    {0x2, HapTextureFormat_RGB_BPTC_UNSIGNED_FLOAT},
    {0x3, HapTextureFormat_RGB_BPTC_SIGNED_FLOAT},
};

unsigned hapf_format_from_nibble(unsigned nibble)
{
    size_t i;
    for (i = 0; i < sizeof(k_formats) / sizeof(k_formats[0]); i++)
        if (k_formats[i].nibble == nibble)
            return k_formats[i].format;
    return 0;
}

unsigned hapf_nibble_from_format(unsigned format)
{
    size_t i;
    for (i = 0; i < sizeof(k_formats) / sizeof(k_formats[0]); i++)
        if (k_formats[i].format == format)
            return k_formats[i].nibble;
    return 0;
}

/* ---------------------------------------------------------- size maths -- */

/* snappy_max_compressed_length of libsnappy: 32 + n + n/6 (call site hap.c:313) */
size_t hapf_snappy_bound(size_t n) { return 32u + n + n / 6u; }

/* compressor table + size table + their two headers (reference hap.c:265-275) */
size_t hapf_instructions_length(unsigned chunks) { return 5u * (size_t)chunks + 8u; }

/* reference hap.c:277-300 */
unsigned hapf_limit_chunk_count(size_t bytes, unsigned format, unsigned chunks)
{
    size_t block = (format == HapTextureFormat_RGB_DXT1 || format == HapTextureFormat_A_RGTC1) ? 8u : 16u;
    unsigned long blocks = (unsigned long)(bytes / block);
    if (chunks > 3355431u)
        chunks = 3355431u;
    while (blocks % chunks)
        chunks--;
    return chunks;
}

/* reference hap.c:302-322 */
size_t hapf_texture_bound(size_t bytes, unsigned format, unsigned compressor, unsigned chunks)
{
    size_t payload = bytes;
    chunks = hapf_limit_chunk_count(bytes, format, chunks);
    if (compressor == HapCompressorSnappy)
        payload = hapf_snappy_bound(bytes / chunks) * chunks;
    return payload + 8u + hapf_instructions_length(chunks) + 4u;
}

/* -------------------------------------------------------------- decode -- */

void hapf_plan_free(hapf_texture_plan *p)
{
    free(p->chunks);
    p->chunks = NULL;
}

/* reference hap.c:932-991 */
unsigned hapf_locate(hapf_reader *r, uint32_t frame_bytes, unsigned index,
                     uint64_t *section_offset, uint32_t *section_length, unsigned *section_type)
{
    hapf_section top, s;
    int rc = read_section_at(r, 0, frame_bytes, &top);
    if (rc != HapResult_No_Error)
        return (unsigned)rc;
    if (top.type == HAP_SECTION_MULTI) {
        uint64_t cursor = 0;
        unsigned i;
        s.header_len = 0;
        s.length = 0;
        for (i = 0; i <= index; i++) {
            cursor += (uint64_t)s.header_len + s.length;
            if (cursor >= top.length)
                return HapResult_Bad_Arguments;
            rc = read_section_at(r, top.header_len + cursor, (uint32_t)(top.length - cursor), &s);
            if (rc != HapResult_No_Error)
                return (unsigned)rc;
        }
        *section_offset = top.header_len + cursor + s.header_len;
        *section_length = s.length;
        *section_type = s.type;
        if (*section_offset + s.length > frame_bytes)
            return HapResult_Bad_Frame;
        return HapResult_No_Error;
    }
    if (index != 0)
        return HapResult_Bad_Arguments;
    *section_offset = top.header_len;
    *section_length = top.length;
    *section_type = top.type;
    if (*section_offset + top.length > frame_bytes)
        return HapResult_Bad_Frame;
    return HapResult_No_Error;
}

/* Walks the Decode Instructions Container (reference hap.c:644-730) and, when
 * asked, lays out the chunk list (reference hap.c:794-809). */
static void plan_complex(hapf_reader *r, hapf_texture_plan *plan, int want_chunks)
{
    hapf_section box, s;
    uint64_t left, at = 0, base;
    int64_t codecs = -1, sizes = -1, offsets = -1, frags = -1;
    uint32_t frag_bytes = 0;
    int rc = read_section_at(r, plan->section_offset, plan->section_length, &box);
    plan->chunk_count = 0;
    if (rc == HapResult_No_Error && box.type != HAP_SECTION_INSTRUCTIONS)
        rc = HapResult_Bad_Frame;
    if (rc != HapResult_No_Error) {
        plan->result = (unsigned)rc;
        return;
    }
    plan->payload_offset = plan->section_offset + box.header_len + box.length;
    plan->payload_length = plan->section_length - (box.header_len + box.length);
    /* The tables are read piece by piece (section headers, then the few tables of interest): the body of
       the fragment-size section can be tens of kilobytes that the host never looks at, and frames in
       device memory only have a short prefix copied back. */
    base = plan->section_offset + box.header_len;
    left = box.length;
    while (left > 0) {
        unsigned n = 0;
        rc = read_section_at(r, base + at, (uint32_t)left, &s);
        if (rc != HapResult_No_Error) {
            plan->result = (unsigned)rc;
            return;
        }
        at += s.header_len;
        switch (s.type) {
        case HAP_SECTION_COMPRESSORS: codecs = (int64_t)at; n = s.length; break;
        case HAP_SECTION_SIZES: sizes = (int64_t)at; n = s.length / 4u; break;
        case HAP_SECTION_OFFSETS: offsets = (int64_t)at; n = s.length / 4u; break;
        case HAP_SECTION_FRAGMENTS: frags = (int64_t)at; frag_bytes = s.length; break;
        default: break;               /* unknown sections are skipped */
        }
        if (n > 0x7FFFFFFFu) {
            plan->result = HapResult_Bad_Frame;
            return;
        }
        if (n != 0) {
            if (plan->chunk_count != 0 && (int)n != plan->chunk_count) {
                plan->result = HapResult_Bad_Frame;
                return;
            }
            plan->chunk_count = (int)n;
        }
        at += s.length;
        left -= (uint64_t)s.header_len + s.length;
    }
    if (codecs < 0 || sizes < 0) {
        plan->result = HapResult_Bad_Frame;
        return;
    }
    if (frags >= 0 && frag_bytes >= 4u && (frag_bytes & 3u) == 0) {
        const uint8_t *fh = hapf_need(r, base + (uint64_t)frags, 4u);
        if (!fh) {
            plan->result = HapResult_Bad_Frame;
            return;
        }
        if (fh[0] == HAP_FRAGMENT_TABLE_VERSION && fh[1] >= 10 && fh[1] <= 16) {
            plan->frag_log2 = fh[1];
            plan->frag_gran_log2 = fh[2] <= 2 ? fh[2] : 0u;
            plan->frag_window256 = fh[3];
            plan->frag_entries = (frag_bytes - 4u) / 4u;
            plan->frag_table_offset = base + (uint64_t)frags + 4u;
        } else if (fh[0] == HAP_FRAGMENT_TABLE_VERSION_FIELDS && fh[1] == 13u && (frag_bytes - 4u) % (4u + HAP_GROUP_TABLE_BYTES) == 0u &&
                   ((fh[2] >> 4) == 4u || (fh[2] >> 4) == 2u || (fh[2] >> 4) == 6u || (fh[2] >> 4) == 8u)) {
            /* version 4: [4][13][granularity log2 | fields per block << 4][window] + u32 x N + 196-byte group table x N */
            plan->frag_log2 = 13u;
            plan->frag_gran_log2 = (fh[2] & 15u) <= 2u ? (fh[2] & 15u) : 0u;
            plan->frag_fields = fh[2] >> 4;
            plan->frag_window256 = fh[3];
            plan->frag_entries = (frag_bytes - 4u) / (4u + HAP_GROUP_TABLE_BYTES);
            plan->frag_table_offset = base + (uint64_t)frags + 4u;
            plan->frag_tiles_offset = plan->frag_table_offset + 4u * (uint64_t)plan->frag_entries;
        } else if (fh[0] == HAP_FRAGMENT_TABLE_VERSION_FIELDS_R4 && fh[1] == 13u &&
                   (frag_bytes - 4u) % (4u + HAP_GROUP_TABLE_BYTES_R4) == 0u) {
            /* version 3 (written until round 4: group tables without the groups' output bytes): the fragment sizes are
               used, the group tables are not -- one wavefront per fragment through the generic kernels */
            plan->frag_log2 = 13u;
            plan->frag_gran_log2 = (fh[2] & 15u) <= 2u ? (fh[2] & 15u) : 0u;
            plan->frag_window256 = fh[3];
            plan->frag_entries = (frag_bytes - 4u) / (4u + HAP_GROUP_TABLE_BYTES_R4);
            plan->frag_table_offset = base + (uint64_t)frags + 4u;
        }
    }
    if (want_chunks && plan->chunk_count > 0) {
        const size_t n = (size_t)plan->chunk_count;
        const uint8_t *t;
        uint64_t run = 0;
        int i;
        plan->chunks = (HapGpuChunkIn *)calloc(n, sizeof(HapGpuChunkIn));
        if (!plan->chunks) {
            plan->result = HapResult_Internal_Error;
            return;
        }
        /* (each table is consumed before the next one is asked for: a fetched range stays valid only until
           the next request) */
        t = hapf_need(r, base + (uint64_t)codecs, n);
        for (i = 0; t && i < plan->chunk_count; i++)
            plan->chunks[i].codec = t[i];
        if (t)
            t = hapf_need(r, base + (uint64_t)sizes, 4u * n);
        for (i = 0; t && i < plan->chunk_count; i++)
            plan->chunks[i].src_len = le32(t + 4 * (size_t)i);
        if (t && offsets >= 0) {
            t = hapf_need(r, base + (uint64_t)offsets, 4u * n);
            for (i = 0; t && i < plan->chunk_count; i++)
                plan->chunks[i].src_off = le32(t + 4 * (size_t)i);
        }
        if (!t) {
            plan->result = HapResult_Bad_Frame;
            return;
        }
        for (i = 0; i < plan->chunk_count; i++) {
            HapGpuChunkIn *c = &plan->chunks[i];
            const uint64_t begin = offsets >= 0 ? c->src_off : run;
            run += c->src_len;
            /* hardening (the reference has no such check, hap.c:798-809) */
            if (begin + c->src_len > plan->payload_length) {
                plan->result = HapResult_Bad_Frame;
                return;
            }
            c->src_off = (uint32_t)begin;
        }
    }
}

void hapf_plan_texture(hapf_reader *r, uint32_t frame_bytes, unsigned index, int want_chunks,
                       hapf_texture_plan *plan)
{
    unsigned type = 0, codec;
    memset(plan, 0, sizeof(*plan));
    plan->mode = 0xFFu;
    plan->result = hapf_locate(r, frame_bytes, index, &plan->section_offset, &plan->section_length, &type);
    if (plan->result != HapResult_No_Error)
        return;
    plan->format = hapf_format_from_nibble(type & 0xFu);
    codec = (type >> 4) & 0xFu;
    if (want_chunks && plan->format == 0) {       /* HapDecode: reference hap.c:748-758 */
        plan->result = HapResult_Bad_Frame;
        return;
    }
    if (codec == HAP_NIBBLE_COMPLEX) {
        plan->mode = HAPGPU_JOB_COMPLEX;
        plan_complex(r, plan, want_chunks);
    } else if (codec == HAP_NIBBLE_SNAPPY) {
        plan->mode = HAPGPU_JOB_SNAPPY;
        plan->chunk_count = 1;
    } else if (codec == HAP_NIBBLE_NONE) {
        plan->mode = HAPGPU_JOB_RAW;
        plan->chunk_count = 1;
    } else {
        plan->result = HapResult_Bad_Frame;       /* reference hap.c:917-920, 1182-1185 */
    }
}

/* reference hap.c:1042-1087, including the offset-vs-length quirk at 1061-1064 */
unsigned hapf_texture_count(hapf_reader *r, unsigned long frame_bytes, unsigned *count)
{
    hapf_section top, s;
    int rc = read_section_at(r, 0, (uint32_t)frame_bytes, &top);
    if (rc != HapResult_No_Error)
        return (unsigned)rc;
    if (top.type == HAP_SECTION_MULTI) {
        uint32_t cursor = top.header_len;
        *count = 0;
        while (cursor < top.length) {
            rc = read_section_at(r, cursor, (uint32_t)(frame_bytes - cursor), &s);
            if (rc != HapResult_No_Error)
                return (unsigned)rc;
            if ((uint64_t)cursor + s.header_len + s.length > 0xFFFFFFFFu)
                return HapResult_Bad_Frame;
            cursor += s.header_len + s.length;
            *count += 1;
        }
        return HapResult_No_Error;
    }
    *count = 1;
    return HapResult_No_Error;
}
