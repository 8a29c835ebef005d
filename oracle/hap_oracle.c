/*
 * hap_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar restatement of the Hap frame container and its chunked Snappy second
 * stage, following /root/reference/source/hap.c.  Each function cites the
 * reference lines whose observable behaviour it restates (including the
 * quirks listed in SURVEY.md App. E).  The Snappy calls go to the restatement
 * in snappy_oracle.c, so this file + snappy_oracle.c are a self-contained CPU
 * model of the whole path; tests pin it against oracle/_ref/libhap_ref.so
 * (the unmodified reference + libsnappy 1.1.8) and tests/golden/.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { R_OK = 0, R_BAD_ARGS = 1, R_TOO_SMALL = 2, R_BAD_FRAME = 3, R_INTERNAL = 4 };
enum { FMT_DXT1 = 0x83F0, FMT_DXT5 = 0x83F3, FMT_YCOCG = 0x01, FMT_RGTC1 = 0x8DBB,
       FMT_BC7 = 0x8E8C, FMT_BC6U = 0x8E8F, FMT_BC6S = 0x8E8E };
enum { COMP_NONE = 0, COMP_SNAPPY = 1 };
enum { NIB_NONE = 0xA, NIB_SNAPPY = 0xB, NIB_COMPLEX = 0xC };
enum { SEC_MULTI = 0x0D, SEC_INSTR = 0x01, SEC_COMPRESSORS = 0x02, SEC_SIZES = 0x03, SEC_OFFSETS = 0x04 };

/* ---- little-endian helpers (hap.c:106-129) ---- */
static uint32_t get24(const uint8_t *p) { return p[0] | (p[1] << 8) | ((uint32_t)p[2] << 16); }
static uint32_t get32(const uint8_t *p) { return get24(p) | ((uint32_t)p[3] << 24); }
static void put24(uint8_t *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; }
static void put32(uint8_t *p, uint32_t v) { put24(p, v); p[3] = (v >> 24) & 255; }

typedef struct { uint32_t hdr, len; unsigned type; } section_t;

/* hap.c:137-187 -- 4-byte header, or 8-byte when the 24-bit length is zero */
static int read_section(const uint8_t *p, uint32_t avail, section_t *s)
{
    if (avail < 4)
        return R_BAD_FRAME;
    s->len = get24(p);
    s->hdr = 4;
    if (s->len == 0) {
        if (avail < 8)
            return R_BAD_FRAME;
        s->len = get32(p + 4);
        s->hdr = 8;
    }
    s->type = p[3];
    /* 32-bit wrap is intentional: the reference adds two uint32_t */
    if ((uint32_t)(s->hdr + s->len) > avail)
        return R_BAD_FRAME;
    return R_OK;
}

/* hap.c:189-212 */
static void write_section(uint8_t *p, size_t hdr, uint32_t len, unsigned type)
{
    if (hdr == 4) {
        put24(p, len);
    } else {
        put24(p, 0);
        put32(p + 4, len);
    }
    p[3] = (uint8_t)type;
}

/* hap.c:215-261 */
static unsigned format_from_nibble(unsigned nib)
{
    switch (nib) {
    case 0xB: return FMT_DXT1;
    case 0xE: return FMT_DXT5;
    case 0xF: return FMT_YCOCG;
    case 0x1: return FMT_RGTC1;
    case 0xC: return FMT_BC7;
    case 0x2: return FMT_BC6U;
    case 0x3: return FMT_BC6S;
    }
    return 0;
}

static unsigned nibble_from_format(unsigned fmt)
{
    switch (fmt) {
    case FMT_DXT1: return 0xB;
    case FMT_DXT5: return 0xE;
    case FMT_YCOCG: return 0xF;
    case FMT_RGTC1: return 0x1;
    case FMT_BC7: return 0xC;
    case FMT_BC6U: return 0x2;
    case FMT_BC6S: return 0x3;
    }
    return 0;
}

/* hap.c:265-275 */
static size_t instr_len(unsigned chunks) { return 5u * (size_t)chunks + 8u; }

/* hap.c:277-300 -- largest divisor of the block count not above the request */
static unsigned limit_chunks(size_t bytes, unsigned fmt, unsigned chunks)
{
    unsigned long blocks;
    if (chunks > 3355431u)
        chunks = 3355431u;
    blocks = (fmt == FMT_DXT1 || fmt == FMT_RGTC1) ? bytes / 8 : bytes / 16;
    while (blocks % chunks != 0)
        chunks--;
    return chunks;
}

/* hap.c:302-322 */
static size_t texture_worst_case(size_t bytes, unsigned fmt, unsigned comp, unsigned chunks)
{
    size_t payload;
    chunks = limit_chunks(bytes, fmt, chunks);
    if (comp == COMP_SNAPPY)
        payload = osnappy_max_compressed_length(bytes / chunks) * chunks;
    else
        payload = bytes;
    return payload + 8u + instr_len(chunks) + 4u;
}

/* hap.c:324-353 */
unsigned long ohap_max_encoded_length(unsigned count, const unsigned long *lengths,
                                      const unsigned *formats, const unsigned *chunk_counts)
{
    unsigned long total = 8;
    unsigned i;
    if (count == 0 || count > 2 || !lengths || !formats || !chunk_counts)
        return 0;
    for (i = 0; i < count; i++) {
        if (chunk_counts[i] == 0)
            return 0;
        total += texture_worst_case(lengths[i], formats[i], COMP_SNAPPY, chunk_counts[i]);
    }
    return total;
}

/* hap.c:355-504 */
static unsigned encode_texture(const uint8_t *in, unsigned long bytes, unsigned fmt, unsigned comp,
                               unsigned chunks, uint8_t *out, unsigned long out_bytes,
                               unsigned long *used)
{
    size_t hdr, body = 0;
    unsigned stored = NIB_NONE;

    if (!in || bytes == 0 || nibble_from_format(fmt) == 0 ||
        (comp != COMP_NONE && comp != COMP_SNAPPY) || !out || !used)
        return R_BAD_ARGS;
    if (out_bytes < texture_worst_case(bytes, fmt, comp, chunks))
        return R_TOO_SMALL;

    hdr = bytes > 0xFFFFFFu ? 8 : 4;                       /* hap.c:398-405 */

    if (comp == COMP_SNAPPY) {
        size_t ilen, chunk_bytes, room;
        uint8_t *ctab, *stab, *dst;
        unsigned i;
        chunks = limit_chunks(bytes, fmt, chunks);
        ilen = instr_len(chunks);
        if (bytes + ilen + 4 > 0xFFFFFFu)                  /* hap.c:425-428 */
            hdr = 8;
        ctab = out + hdr + 8;
        stab = ctab + chunks + 4;
        chunk_bytes = bytes / chunks;
        write_section(out + hdr, 4, (uint32_t)ilen, SEC_INSTR);
        write_section(out + hdr + 4, 4, chunks, SEC_COMPRESSORS);
        write_section(out + hdr + 8 + chunks, 4, chunks * 4u, SEC_SIZES);
        dst = out + hdr + 4 + ilen;
        room = out_bytes - hdr - 4 - ilen;
        body = 4 + ilen;
        for (i = 0; i < chunks; i++) {                     /* hap.c:448-476 */
            const uint8_t *src = in + chunk_bytes * i;
            size_t packed = room;
            if (osnappy_compress(src, chunk_bytes, dst, &packed) != OSNAPPY_OK)
                return R_INTERNAL;
            if (packed >= chunk_bytes) {
                memcpy(dst, src, chunk_bytes);
                packed = chunk_bytes;
                ctab[i] = NIB_NONE;
            } else {
                ctab[i] = NIB_SNAPPY;
            }
            put32(stab + 4 * i, (uint32_t)packed);
            dst += packed;
            body += packed;
            room -= packed;
        }
        if (body < bytes + hdr)                            /* hap.c:478-487 */
            stored = NIB_COMPLEX;
        else
            comp = COMP_NONE;
    }
    if (comp == COMP_NONE) {                               /* hap.c:490-495 */
        memcpy(out + hdr, in, bytes);
        body = bytes;
        stored = NIB_NONE;
    }
    write_section(out, hdr, (uint32_t)body, (stored << 4) | (nibble_from_format(fmt) & 0xF));
    *used = body + hdr;
    return R_OK;
}

/* hap.c:506-604 */
unsigned ohap_encode(unsigned count, const void **inputs, const unsigned long *input_bytes,
                     const unsigned *formats, const unsigned *compressors,
                     const unsigned *chunk_counts, void *out, unsigned long out_bytes,
                     unsigned long *out_used)
{
    unsigned i;
    size_t hdr, body;
    if (count == 0 || count > 2 || !inputs || !input_bytes || !formats || !compressors ||
        !chunk_counts || !out || out_bytes == 0 || !out_used)
        return R_BAD_ARGS;
    for (i = 0; i < count; i++)
        if (chunk_counts[i] == 0)
            return R_BAD_ARGS;
    if (count == 1)
        return encode_texture(inputs[0], input_bytes[0], formats[0], compressors[0],
                              chunk_counts[0], out, out_bytes, out_used);
    /* permissive pair check, hap.c:551-552 */
    if (formats[0] != FMT_YCOCG && formats[1] != FMT_YCOCG &&
        formats[0] != FMT_RGTC1 && formats[1] != FMT_RGTC1)
        return R_BAD_ARGS;
    body = 0;
    for (i = 0; i < count; i++)                            /* hap.c:563-576: un-limited count */
        body += input_bytes[i] + instr_len(chunk_counts[i]) + 4;
    hdr = body > 0xFFFFFFu ? 8 : 4;
    body = 0;
    for (i = 0; i < count; i++) {
        unsigned long one = 0;
        unsigned r = encode_texture(inputs[i], input_bytes[i], formats[i], compressors[i],
                                    chunk_counts[i], (uint8_t *)out + hdr + body,
                                    out_bytes - (hdr + body), &one);
        if (r != R_OK)
            return r;
        body += one;
    }
    write_section(out, hdr, (uint32_t)body, SEC_MULTI);
    *out_used = body + hdr;
    return R_OK;
}

/* ---- decode ---- */

typedef struct {
    unsigned result, codec;
    const uint8_t *src;
    size_t src_len;
    uint8_t *dst;
    size_t dst_len;
} chunk_job;

/* hap.c:606-642 */
static void run_chunk(void *p, unsigned i)
{
    chunk_job *j = (chunk_job *)p;
    if (!j)
        return;
    j += i;
    if (j->codec == NIB_SNAPPY) {
        int s = osnappy_uncompress(j->src, j->src_len, j->dst, &j->dst_len);
        j->result = s == OSNAPPY_OK ? R_OK : s == OSNAPPY_INVALID_INPUT ? R_BAD_FRAME : R_INTERNAL;
    } else if (j->codec == NIB_NONE) {
        memcpy(j->dst, j->src, j->src_len);
        j->result = R_OK;
    } else {
        j->result = R_BAD_FRAME;
    }
}

typedef struct { int chunks; const uint8_t *codecs, *sizes, *offsets, *payload; } instr_t;

/* hap.c:644-730 */
static int parse_instructions(const uint8_t *sec, uint32_t sec_len, instr_t *t)
{
    section_t s;
    const uint8_t *p;
    size_t left;
    int r;
    t->codecs = t->sizes = t->offsets = NULL;
    r = read_section(sec, sec_len, &s);
    if (r == R_OK && s.type != SEC_INSTR)
        r = R_BAD_FRAME;
    if (r != R_OK)
        return r;
    t->payload = sec + s.hdr + s.len;
    p = sec + s.hdr;
    left = s.len;
    while (left > 0) {
        unsigned n = 0;
        r = read_section(p, (uint32_t)left, &s);
        if (r != R_OK)
            return r;
        p += s.hdr;
        if (s.type == SEC_COMPRESSORS) { t->codecs = p; n = s.len; }
        else if (s.type == SEC_SIZES) { t->sizes = p; n = s.len / 4; }
        else if (s.type == SEC_OFFSETS) { t->offsets = p; n = s.len / 4; }
        if (n != 0) {
            if (t->chunks != 0 && (int)n != t->chunks)
                return R_BAD_FRAME;
            t->chunks = (int)n;
        }
        p += s.len;
        left -= s.hdr + s.len;
    }
    if (!t->codecs || !t->sizes)
        return R_BAD_FRAME;
    return R_OK;
}

/* hap.c:932-991 */
static int locate_texture(const uint8_t *in, uint32_t in_bytes, unsigned index,
                          const uint8_t **sec, uint32_t *sec_len, unsigned *type)
{
    section_t s;
    int r = read_section(in, in_bytes, &s);
    if (r != R_OK)
        return r;
    *sec_len = s.len;
    *type = s.type;
    if (s.type == SEC_MULTI) {
        size_t off = 0, top = s.len;
        unsigned i;
        const uint8_t *body = in + s.hdr;
        s.hdr = 0;
        s.len = 0;
        *sec_len = 0;
        for (i = 0; i <= index; i++) {
            off += s.hdr + s.len;
            if (off >= top)
                return R_BAD_ARGS;
            r = read_section(body + off, (uint32_t)(top - off), &s);
            *sec_len = s.len;     /* the reference writes outputs before checking */
            *type = s.type;
            if (r != R_OK)
                return r;
        }
        *sec = body + off + s.hdr;
        return R_OK;
    }
    if (index == 0) {
        *sec = in + s.hdr;
        return R_OK;
    }
    *sec = NULL;
    *sec_len = 0;
    *type = 0;
    return R_BAD_ARGS;
}

/* hap.c:732-930 */
static unsigned decode_texture(const uint8_t *sec, uint32_t sec_len, unsigned type, OHapCallback cb,
                               void *info, uint8_t *out, unsigned long out_bytes,
                               unsigned long *used, unsigned *fmt)
{
    unsigned codec = (type >> 4) & 0xF;
    size_t produced = 0;
    *fmt = format_from_nibble(type & 0xF);
    if (*fmt == 0)
        return R_BAD_FRAME;
    if (codec == NIB_COMPLEX) {
        instr_t t;
        int r, i;
        t.chunks = 0;
        r = parse_instructions(sec, sec_len, &t);
        if (r != R_OK)
            return (unsigned)r;
        if (t.chunks > 0) {
            chunk_job *jobs = (chunk_job *)malloc(sizeof(chunk_job) * (size_t)t.chunks);
            size_t in_run = 0, out_run = 0;
            if (!jobs)
                return R_INTERNAL;
            for (i = 0; i < t.chunks; i++) {               /* hap.c:794-838 */
                jobs[i].codec = t.codecs[i];
                jobs[i].src_len = get32(t.sizes + 4 * i);
                jobs[i].src = t.payload + (t.offsets ? get32(t.offsets + 4 * i) : in_run);
                in_run += jobs[i].src_len;
                if (jobs[i].codec == NIB_SNAPPY) {
                    int s = osnappy_uncompressed_length(jobs[i].src, jobs[i].src_len, &jobs[i].dst_len);
                    if (s != OSNAPPY_OK) {
                        r = s == OSNAPPY_INVALID_INPUT ? R_BAD_FRAME : R_INTERNAL;
                        break;
                    }
                } else {
                    jobs[i].dst_len = jobs[i].src_len;
                }
                jobs[i].dst = out + out_run;
                out_run += jobs[i].dst_len;
            }
            if (r == R_OK && out_run > out_bytes)
                r = R_TOO_SMALL;
            if (r == R_OK) {
                produced = out_run;
                if (t.chunks == 1)
                    run_chunk(jobs, 0);
                else
                    cb(run_chunk, jobs, (unsigned)t.chunks, info);
                for (i = 0; i < t.chunks; i++)
                    if (jobs[i].result != R_OK) {
                        r = (int)jobs[i].result;
                        break;
                    }
            }
            free(jobs);
            if (r != R_OK)
                return (unsigned)r;
        }
    } else if (codec == NIB_SNAPPY) {                      /* hap.c:885-904 */
        if (osnappy_uncompressed_length(sec, sec_len, &produced) != OSNAPPY_OK)
            return R_INTERNAL;
        if (produced > out_bytes)
            return R_TOO_SMALL;
        if (osnappy_uncompress(sec, sec_len, out, &produced) != OSNAPPY_OK)
            return R_INTERNAL;
    } else if (codec == NIB_NONE) {                        /* hap.c:905-916 */
        produced = sec_len;
        if (sec_len > out_bytes)
            return R_TOO_SMALL;
        memcpy(out, sec, sec_len);
    } else {
        return R_BAD_FRAME;
    }
    if (used)
        *used = produced;
    return R_OK;
}

/* hap.c:993-1040 */
unsigned ohap_decode(const void *in, unsigned long in_bytes, unsigned index, OHapCallback cb,
                     void *info, void *out, unsigned long out_bytes, unsigned long *out_used,
                     unsigned *out_format)
{
    const uint8_t *sec;
    uint32_t sec_len;
    unsigned type;
    int r;
    if (!in || index > 1 || !cb || !out || !out_format)
        return R_BAD_ARGS;
    r = locate_texture((const uint8_t *)in, (uint32_t)in_bytes, index, &sec, &sec_len, &type);
    if (r != R_OK)
        return (unsigned)r;
    return decode_texture(sec, sec_len, type, cb, info, (uint8_t *)out, out_bytes, out_used, out_format);
}

/* hap.c:1042-1087 */
unsigned ohap_texture_count(const void *in, unsigned long in_bytes, unsigned *count)
{
    section_t s;
    const uint8_t *p = (const uint8_t *)in;
    int r = read_section(p, (uint32_t)in_bytes, &s);
    if (r != R_OK)
        return (unsigned)r;
    if (s.type == SEC_MULTI) {
        uint32_t off = s.hdr, top = s.len;     /* quirk: offset includes the header, top does not */
        *count = 0;
        while (off < top) {
            r = read_section(p + off, (uint32_t)(in_bytes - off), &s);
            if (r != R_OK)
                return (unsigned)r;
            off += s.hdr + s.len;
            *count += 1;
        }
        return R_OK;
    }
    *count = 1;
    return R_OK;
}

/* hap.c:1089-1126 */
unsigned ohap_texture_format(const void *in, unsigned long in_bytes, unsigned index, unsigned *fmt)
{
    const uint8_t *sec;
    uint32_t sec_len;
    unsigned type;
    int r;
    if (!in || index > 1 || !fmt)
        return R_BAD_ARGS;
    r = locate_texture((const uint8_t *)in, (uint32_t)in_bytes, index, &sec, &sec_len, &type);
    if (r != R_OK)
        return (unsigned)r;
    *fmt = format_from_nibble(type & 0xF);
    return *fmt ? R_OK : R_BAD_FRAME;
}

/* hap.c:1128-1188 */
unsigned ohap_texture_chunk_count(const void *in, unsigned long in_bytes, unsigned index, int *n)
{
    const uint8_t *sec;
    uint32_t sec_len;
    unsigned type, codec;
    int r;
    *n = 0;                                    /* written before validation, hap.c:1134 */
    if (!in || index > 1)
        return R_BAD_ARGS;
    r = locate_texture((const uint8_t *)in, (uint32_t)in_bytes, index, &sec, &sec_len, &type);
    if (r != R_OK)
        return (unsigned)r;
    codec = (type >> 4) & 0xF;
    if (codec == NIB_COMPLEX) {
        instr_t t;
        t.chunks = 0;
        r = parse_instructions(sec, sec_len, &t);
        *n = t.chunks;
        return (unsigned)r;
    }
    if (codec == NIB_SNAPPY || codec == NIB_NONE) {
        *n = 1;
        return R_OK;
    }
    return R_BAD_FRAME;
}
