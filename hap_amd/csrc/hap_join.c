/*
 * hap_join.c -- joins frames that each carry one contiguous group of a
 * texture's chunks into one Hap frame (one huge frame split
 * over GPUs by chunk groups; every GPU encodes its band of block rows as a
 * frame of its own, the root joins them).  Host-only scalar work on the
 * tables; payload bytes are copied once.
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
    unsigned frag_log2, frag_gran_log2, frag_window256, frags_per_chunk;
    uint64_t payload;         /* stored bytes of all chunks */
    uint64_t body;            /* section length, header excluded */
    unsigned header_len;
} joined_texture;

static uint64_t instructions_bytes(const joined_texture *t)
{
    uint64_t n = hapf_instructions_length(t->chunk_count);
    if (t->keep_index)
        n += 8u + 4u * (uint64_t)t->chunk_count * t->frags_per_chunk;
    return n;
}

unsigned int HapGpuJoinChunkGroups(unsigned int groupCount, const void *const *groupFrames,
                                   const unsigned long *groupFramesBytes, void *outputBuffer,
                                   unsigned long outputBufferBytes, unsigned long *outputBufferBytesUsed)
{
    hapf_reader *readers = NULL;
    hapf_texture_plan *plans = NULL;   /* [texture][group] */
    joined_texture tex[2];
    unsigned count = 0, g, t, result = HapResult_No_Error;
    uint64_t total = 0, outer_header = 0;
    uint8_t *out = (uint8_t *)outputBuffer, *cursor;

    if (groupCount == 0 || !groupFrames || !groupFramesBytes || !outputBuffer || !outputBufferBytesUsed)
        return HapResult_Bad_Arguments;
    for (g = 0; g < groupCount; g++)
        if (!groupFrames[g] || groupFramesBytes[g] > 0xFFFFFFFFul)
            return HapResult_Bad_Arguments;

    readers = (hapf_reader *)calloc(groupCount, sizeof(*readers));
    plans = (hapf_texture_plan *)calloc((size_t)groupCount * 2u, sizeof(*plans));
    if (!readers || !plans) {
        free(readers); free(plans);
        return HapResult_Internal_Error;
    }
    for (g = 0; g < groupCount; g++) {
        unsigned n = 0, r;
        hapf_reader_init_host(&readers[g], groupFrames[g], groupFramesBytes[g]);
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
                } else if (per_chunk != j->frags_per_chunk || p->frag_log2 != j->frag_log2) {
                    j->keep_index = 0;
                }
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

    cursor = out + outer_header;
    for (t = 0; t < count; t++) {
        const joined_texture *j = &tex[t];
        uint8_t *sec = cursor, *ctab = NULL, *stab = NULL, *ftab = NULL, *payload;
        unsigned chunk = 0;
        if (j->all_raw) {
            hapf_write_section(sec, j->header_len, (uint32_t)j->body, (HAP_NIBBLE_NONE << 4) | j->format_nibble);
            payload = sec + j->header_len;
        } else {
            const uint32_t ilen = (uint32_t)instructions_bytes(j);
            const unsigned n = j->chunk_count;
            hapf_write_section(sec, j->header_len, (uint32_t)j->body, (HAP_NIBBLE_COMPLEX << 4) | j->format_nibble);
            hapf_write_section(sec + j->header_len, 4u, ilen, HAP_SECTION_INSTRUCTIONS);       /* hap.c:436 */
            hapf_write_section(sec + j->header_len + 4u, 4u, n, HAP_SECTION_COMPRESSORS);      /* hap.c:438 */
            ctab = sec + j->header_len + 8u;
            hapf_write_section(ctab + n, 4u, 4u * n, HAP_SECTION_SIZES);                       /* hap.c:440 */
            stab = ctab + n + 4u;
            if (j->keep_index) {
                uint8_t *isec = stab + 4u * (size_t)n;
                hapf_write_section(isec, 4u, 4u + 4u * n * j->frags_per_chunk, HAP_SECTION_FRAGMENTS);
                isec[4] = (uint8_t)HAP_FRAGMENT_TABLE_VERSION;
                isec[5] = (uint8_t)j->frag_log2;
                isec[6] = (uint8_t)j->frag_gran_log2;
                isec[7] = (uint8_t)j->frag_window256;
                ftab = isec + 8u;
            }
            payload = sec + j->header_len + 4u + ilen;
        }
        for (g = 0; g < groupCount; g++) {
            const hapf_texture_plan *p = &plans[t * groupCount + g];
            const uint8_t *frame = (const uint8_t *)groupFrames[g];
            int c;
            if (p->mode != HAPGPU_JOB_COMPLEX) {
                memcpy(payload, frame + p->section_offset, p->section_length);
                if (!j->all_raw) {
                    ctab[chunk] = p->mode == HAPGPU_JOB_RAW ? (uint8_t)HAP_NIBBLE_NONE : (uint8_t)HAP_NIBBLE_SNAPPY;
                    stab[4u * chunk + 0] = (uint8_t)(p->section_length);
                    stab[4u * chunk + 1] = (uint8_t)(p->section_length >> 8);
                    stab[4u * chunk + 2] = (uint8_t)(p->section_length >> 16);
                    stab[4u * chunk + 3] = (uint8_t)(p->section_length >> 24);
                }
                payload += p->section_length;
                chunk++;
                continue;
            }
            for (c = 0; c < p->chunk_count; c++) {
                const uint32_t len = p->chunks[c].src_len;
                memcpy(payload, frame + p->payload_offset + p->chunks[c].src_off, len);
                payload += len;
                if (!j->all_raw) {
                    ctab[chunk] = (uint8_t)(p->chunks[c].codec & 0xFFu);
                    stab[4u * chunk + 0] = (uint8_t)(len);
                    stab[4u * chunk + 1] = (uint8_t)(len >> 8);
                    stab[4u * chunk + 2] = (uint8_t)(len >> 16);
                    stab[4u * chunk + 3] = (uint8_t)(len >> 24);
                }
                chunk++;
            }
            if (ftab) {
                memcpy(ftab, frame + p->frag_table_offset, 4u * (size_t)p->frag_entries);
                ftab += 4u * (size_t)p->frag_entries;
            }
        }
        cursor += j->header_len + j->body;
    }
    if (outer_header)
        hapf_write_section(out, (unsigned)outer_header, (uint32_t)total, HAP_SECTION_MULTI);   /* hap.c:598 */
    *outputBufferBytesUsed = (unsigned long)(outer_header + total);

done:
    for (g = 0; g < groupCount; g++) {
        for (t = 0; t < 2; t++)
            hapf_plan_free(&plans[t * groupCount + g]);
        hapf_reader_free(&readers[g]);
    }
    free(readers);
    free(plans);
    return result;
}
