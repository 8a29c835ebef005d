/*
 * hap_join.c -- joins frames that each carry one contiguous group of a
 * texture's chunks into one Hap frame (one huge frame split
 * over GPUs by chunk groups; every GPU encodes its band of block rows as a
 * frame of its own, the root joins them).  Scalar work on the tables; payload
 * bytes move once, through a sink: memcpy for frames in host memory
 * (HapGpuJoinChunkGroups), device-to-device moves for frames in HBM
 * (HapGpuJoinChunkGroupsDevice in hap_batch.c: the band frames of a node's
 * GPUs arrive over xGMI and never touch host memory).
 *
 * The joined frame is an ordinary Hap frame (layout as written by reference
 * hap.c:430-442 for one texture, hap.c:562-598 for two): its chunk list is
 * the concatenation of the groups' chunk lists.  A group whose section was
 * stored as-is (0xA_, hap.c:490-495) contributes one uncompressed chunk, one
 * stored as a bare Snappy stream (0xB_) one Snappy chunk.  When every chunk
 * ends up uncompressed the texture is written as a plain 0xA_ section, the
 * form the reference picks when compression gains nothing (hap.c:478-495).
 */
#include <stdlib.h>
#include <string.h>

#include "hap_batch.h"

typedef struct joined_texture {
    unsigned format_nibble;
    unsigned chunk_count;
    int all_raw;
    int keep_index;           /* every group brought a compatible fragment table */
    int keep_tiles;           /* ... of version 3 with the same block layout: the group tables are carried over */
    unsigned frag_log2, frag_gran_log2, frag_window256, frags_per_chunk, frag_fields;
    uint64_t payload;         /* stored bytes of all chunks */
    uint64_t body;            /* section length, header excluded */
    unsigned header_len;
} joined_texture;

static uint64_t instructions_bytes(const joined_texture *t)
{
    uint64_t n = hapf_instructions_length(t->chunk_count);
    if (t->keep_index)
        n += 8u + (t->keep_tiles ? 4u + HAP_GROUP_TABLE_BYTES : 4u) * (uint64_t)t->chunk_count * t->frags_per_chunk;
    return n;
}

unsigned hapj_join(unsigned groupCount, hapf_reader *readers, const unsigned long *groupFramesBytes,
                   const hapj_sink *sink, unsigned long outputBufferBytes, unsigned long *outputBufferBytesUsed)
{
    hapf_texture_plan *plans = NULL;   /* [texture][group] */
    joined_texture tex[2];
    unsigned count = 0, g, t, result = HapResult_No_Error;
    uint64_t total = 0, outer_header = 0, cursor;
    uint8_t *hdr = NULL;

    plans = (hapf_texture_plan *)calloc((size_t)groupCount * 2u, sizeof(*plans));
    if (!plans)
        return HapResult_Internal_Error;
    for (g = 0; g < groupCount; g++) {
        unsigned n = 0, r;
        r = hapf_texture_count(&readers[g], groupFramesBytes[g], &n);
        if (r != HapResult_No_Error || n == 0 || n > 2 || (g && n != count)) {
            result = r != HapResult_No_Error ? r : HapResult_Bad_Frame;
            goto done;
        }
        count = n;
    }

    memset(tex, 0, sizeof(tex));
    for (t = 0; t < count; t++) {
        joined_texture *j = &tex[t];
        j->all_raw = 1;
        j->keep_index = 1;
        j->keep_tiles = 1;
        for (g = 0; g < groupCount; g++) {
            hapf_texture_plan *p = &plans[t * groupCount + g];
            unsigned nibble;
            int c;
            hapf_plan_texture(&readers[g], (uint32_t)groupFramesBytes[g], t, 1, p);
            if (p->result != HapResult_No_Error) {
                result = p->result;
                goto done;
            }
            nibble = hapf_nibble_from_format(p->format);
            if (g && nibble != j->format_nibble) {
                result = HapResult_Bad_Frame;   /* groups disagree about the texture format */
                goto done;
            }
            j->format_nibble = nibble;
            if (p->mode != HAPGPU_JOB_COMPLEX) {
                j->chunk_count += 1u;
                j->payload += p->section_length;
                j->keep_index = 0;
                if (p->mode != HAPGPU_JOB_RAW)
                    j->all_raw = 0;
                continue;
            }
            for (c = 0; c < p->chunk_count; c++) {
                if ((uint64_t)p->chunks[c].src_off + p->chunks[c].src_len > p->payload_length) {
                    result = HapResult_Bad_Frame;
                    goto done;
                }
                j->payload += p->chunks[c].src_len;
                if ((p->chunks[c].codec & 0xFFu) != HAP_NIBBLE_NONE)
                    j->all_raw = 0;
            }
            j->chunk_count += (unsigned)p->chunk_count;
            if (!p->frag_table_offset || p->chunk_count <= 0 || p->frag_entries == 0 ||
                p->frag_entries % (unsigned)p->chunk_count) {
                j->keep_index = 0;
            } else {
                const unsigned per_chunk = p->frag_entries / (unsigned)p->chunk_count;
                if (j->frags_per_chunk == 0) {
                    j->frags_per_chunk = per_chunk;
                    j->frag_log2 = p->frag_log2;
                    j->frag_gran_log2 = p->frag_gran_log2;
                    j->frag_window256 = p->frag_window256;
                    j->frag_fields = p->frag_fields;
                } else if (per_chunk != j->frags_per_chunk || p->frag_log2 != j->frag_log2) {
                    j->keep_index = 0;
                }
                if (!p->frag_tiles_offset || !p->frag_fields || p->frag_fields != j->frag_fields)
                    j->keep_tiles = 0;
                if (p->frag_gran_log2 < j->frag_gran_log2)
                    j->frag_gran_log2 = p->frag_gran_log2;   /* the weakest promise holds for all */
                if (p->frag_window256 == 0 || j->frag_window256 == 0)
                    j->frag_window256 = 0;                   /* someone makes no promise: none for the whole */
                else if (p->frag_window256 > j->frag_window256)
                    j->frag_window256 = p->frag_window256;
            }
        }
        if (j->frags_per_chunk == 0)
            j->keep_index = 0;
        if (!j->keep_index)
            j->keep_tiles = 0;
        if (j->all_raw) {
            j->body = j->payload;
        } else {
            const uint64_t ilen = instructions_bytes(j);
            if (ilen + 4u > 0xFFFFFFu) {
                result = HapResult_Bad_Arguments;   /* instruction container must fit a 24-bit length */
                goto done;
            }
            j->body = 4u + ilen + j->payload;
        }
        if (j->body > 0xFFFFFFFFull) {
            result = HapResult_Bad_Arguments;
            goto done;
        }
        j->header_len = j->body > 0xFFFFFFu ? 8u : 4u;             /* hap.c:398-405, 425-428 */
        total += j->header_len + j->body;
    }
    if (count == 2) {
        if (total > 0xFFFFFFFFull) {
            result = HapResult_Bad_Arguments;
            goto done;
        }
        outer_header = total > 0xFFFFFFu ? 8u : 4u;                /* hap.c:562-576 */
    }
    if (outer_header + total > outputBufferBytes) {
        result = HapResult_Buffer_Too_Small;
        goto done;
    }

    /* per texture: the header region (section headers, codec and size tables, fragment-table header) is built here
       and put first; the groups' table contents and payloads are then moved behind it */
    cursor = outer_header;
    for (t = 0; t < count && result == HapResult_No_Error; t++) {
        const joined_texture *j = &tex[t];
        uint64_t payload, ftab = 0, ttab = 0;
        unsigned chunk = 0;
        size_t hdr_len;
        uint8_t *ctab = NULL, *stab = NULL;
        int bad = 0;
        if (j->all_raw) {
            hdr_len = j->header_len;
            hdr = (uint8_t *)calloc(1, hdr_len);
            if (!hdr) { result = HapResult_Internal_Error; break; }
            hapf_write_section(hdr, j->header_len, (uint32_t)j->body, (HAP_NIBBLE_NONE << 4) | j->format_nibble);
            payload = cursor + j->header_len;
        } else {
            const uint32_t ilen = (uint32_t)instructions_bytes(j);
            const unsigned n = j->chunk_count;
            /* (only what is made here: the fragment table's entries behind its 8 header bytes are moved from the groups) */
            hdr_len = (size_t)j->header_len + 4u + 5u * (size_t)n + 8u + (j->keep_index ? 8u : 0u);
            hdr = (uint8_t *)calloc(1, hdr_len);
            if (!hdr) { result = HapResult_Internal_Error; break; }
            hapf_write_section(hdr, j->header_len, (uint32_t)j->body, (HAP_NIBBLE_COMPLEX << 4) | j->format_nibble);
            hapf_write_section(hdr + j->header_len, 4u, ilen, HAP_SECTION_INSTRUCTIONS);       /* hap.c:436 */
            hapf_write_section(hdr + j->header_len + 4u, 4u, n, HAP_SECTION_COMPRESSORS);      /* hap.c:438 */
            ctab = hdr + j->header_len + 8u;
            hapf_write_section(ctab + n, 4u, 4u * n, HAP_SECTION_SIZES);                       /* hap.c:440 */
            stab = ctab + n + 4u;
            if (j->keep_index) {
                uint8_t *isec = stab + 4u * (size_t)n;
                const uint32_t entries = n * j->frags_per_chunk;
                hapf_write_section(isec, 4u, 4u + (j->keep_tiles ? 4u + HAP_GROUP_TABLE_BYTES : 4u) * entries, HAP_SECTION_FRAGMENTS);
                isec[4] = (uint8_t)(j->keep_tiles ? HAP_FRAGMENT_TABLE_VERSION_FIELDS : HAP_FRAGMENT_TABLE_VERSION);
                isec[5] = (uint8_t)j->frag_log2;
                isec[6] = (uint8_t)(j->frag_gran_log2 | (j->keep_tiles ? j->frag_fields << 4 : 0u));
                isec[7] = (uint8_t)j->frag_window256;
                ftab = cursor + (uint64_t)(isec + 8u - hdr);
                ttab = ftab + 4u * (uint64_t)entries;
            }
            payload = cursor + j->header_len + 4u + ilen;
        }
        /* chunk tables first (host bytes), then the moves */
        for (g = 0; g < groupCount && !j->all_raw; g++) {
            const hapf_texture_plan *p = &plans[t * groupCount + g];
            int c;
            if (p->mode != HAPGPU_JOB_COMPLEX) {
                ctab[chunk] = p->mode == HAPGPU_JOB_RAW ? (uint8_t)HAP_NIBBLE_NONE : (uint8_t)HAP_NIBBLE_SNAPPY;
                stab[4u * chunk + 0] = (uint8_t)(p->section_length);
                stab[4u * chunk + 1] = (uint8_t)(p->section_length >> 8);
                stab[4u * chunk + 2] = (uint8_t)(p->section_length >> 16);
                stab[4u * chunk + 3] = (uint8_t)(p->section_length >> 24);
                chunk++;
                continue;
            }
            for (c = 0; c < p->chunk_count; c++) {
                const uint32_t len = p->chunks[c].src_len;
                ctab[chunk] = (uint8_t)(p->chunks[c].codec & 0xFFu);
                stab[4u * chunk + 0] = (uint8_t)(len);
                stab[4u * chunk + 1] = (uint8_t)(len >> 8);
                stab[4u * chunk + 2] = (uint8_t)(len >> 16);
                stab[4u * chunk + 3] = (uint8_t)(len >> 24);
                chunk++;
            }
        }
        bad |= sink->put(sink->user, cursor, hdr, hdr_len);
        free(hdr);
        hdr = NULL;
        for (g = 0; g < groupCount && !bad; g++) {
            const hapf_texture_plan *p = &plans[t * groupCount + g];
            int c;
            if (p->mode != HAPGPU_JOB_COMPLEX) {
                bad |= sink->move(sink->user, g, p->section_offset, payload, p->section_length);
                payload += p->section_length;
                continue;
            }
            /* (the chunks of a group lie back to back in its frame: one move for all of them when they do) */
            for (c = 0; c < p->chunk_count && !bad; c++) {
                const uint32_t len = p->chunks[c].src_len;
                bad |= sink->move(sink->user, g, p->payload_offset + p->chunks[c].src_off, payload, len);
                payload += len;
            }
            if (ftab) {
                bad |= sink->move(sink->user, g, p->frag_table_offset, ftab, 4u * (size_t)p->frag_entries);
                ftab += 4u * (uint64_t)p->frag_entries;
                if (j->keep_tiles) {
                    bad |= sink->move(sink->user, g, p->frag_tiles_offset, ttab, (size_t)HAP_GROUP_TABLE_BYTES * p->frag_entries);
                    ttab += (uint64_t)HAP_GROUP_TABLE_BYTES * p->frag_entries;
                }
            }
        }
        if (bad)
            result = HapResult_Internal_Error;
        cursor += j->header_len + j->body;
    }
    if (outer_header && result == HapResult_No_Error) {
        uint8_t top[8];
        hapf_write_section(top, (unsigned)outer_header, (uint32_t)total, HAP_SECTION_MULTI);   /* hap.c:598 */
        if (sink->put(sink->user, 0, top, (size_t)outer_header))
            result = HapResult_Internal_Error;
    }
    if (result == HapResult_No_Error)
        *outputBufferBytesUsed = (unsigned long)(outer_header + total);

done:
    for (g = 0; g < groupCount; g++)
        for (t = 0; t < 2; t++)
            hapf_plan_free(&plans[t * groupCount + g]);
    free(plans);
    free(hdr);
    return result;
}

/* ---- frames in host memory ---- */
typedef struct host_sink {
    uint8_t *out;
    const void *const *frames;
} host_sink;

static int host_put(void *user, uint64_t dst_off, const void *src, size_t len)
{
    memcpy(((host_sink *)user)->out + dst_off, src, len);
    return 0;
}

static int host_move(void *user, unsigned group, uint64_t src_off, uint64_t dst_off, size_t len)
{
    host_sink *h = (host_sink *)user;
    memcpy(h->out + dst_off, (const uint8_t *)h->frames[group] + src_off, len);
    return 0;
}

unsigned int HapGpuJoinChunkGroups(unsigned int groupCount, const void *const *groupFrames,
                                   const unsigned long *groupFramesBytes, void *outputBuffer,
                                   unsigned long outputBufferBytes, unsigned long *outputBufferBytesUsed)
{
    hapf_reader *readers;
    host_sink hs;
    hapj_sink sink;
    unsigned g, result;
    if (groupCount == 0 || !groupFrames || !groupFramesBytes || !outputBuffer || !outputBufferBytesUsed)
        return HapResult_Bad_Arguments;
    for (g = 0; g < groupCount; g++)
        if (!groupFrames[g] || groupFramesBytes[g] > 0xFFFFFFFFul)
            return HapResult_Bad_Arguments;
    readers = (hapf_reader *)calloc(groupCount, sizeof(*readers));
    if (!readers)
        return HapResult_Internal_Error;
    for (g = 0; g < groupCount; g++)
        hapf_reader_init_host(&readers[g], groupFrames[g], groupFramesBytes[g]);
    hs.out = (uint8_t *)outputBuffer;
    hs.frames = groupFrames;
    sink.user = &hs;
    sink.put = host_put;
    sink.move = host_move;
    result = hapj_join(groupCount, readers, groupFramesBytes, &sink, outputBufferBytes, outputBufferBytesUsed);
    for (g = 0; g < groupCount; g++)
        hapf_reader_free(&readers[g]);
    free(readers);
    return result;
}
