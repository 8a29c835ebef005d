/*
 * hap_frame.h -- host-side Hap container logic (pure C, no HIP): section
 * headers, size arithmetic and the decode planner that turns a frame's tables
 * into the chunk list the GPU consumes.  Microseconds of scalar integer work
 * per frame; it never touches texture payload bytes.
 */
#ifndef HAP_FRAME_H
#define HAP_FRAME_H

#include "hapgpu_abi.h"

typedef struct hapf_section {
    uint32_t header_len; /* 4 or 8 */
    uint32_t length;     /* payload bytes, header excluded */
    unsigned type;
} hapf_section;

/* Byte access to a frame that may live in device memory: `view` is a
 * host-visible copy of the first view_len bytes; anything beyond is pulled
 * through `fetch` on demand. */
typedef struct hapf_reader {
    const uint8_t *view;     /* host-visible bytes from the start of the frame */
    uint64_t view_len;
    const uint8_t *view2;    /* optional second host-visible window (a later texture's section, fetched ahead) */
    uint64_t view2_off, view2_len;
    int (*fetch)(void *user, uint64_t offset, uint64_t length, uint8_t *dst);
    void *user;
    uint64_t total_len;      /* frame length when known (lets a fetch bring a few KiB around what was asked for), else 0 */
    uint8_t *side;           /* last fetched range */
    uint64_t side_cap, side_off, side_len;
} hapf_reader;

void hapf_reader_init_host(hapf_reader *r, const void *frame, uint64_t length);
void hapf_reader_free(hapf_reader *r);
/* pointer to `length` host-visible bytes at `offset`, valid until the next call; NULL on failure */
const uint8_t *hapf_need(hapf_reader *r, uint64_t offset, uint64_t length);

/* What the planner found out about one texture of a frame. */
typedef struct hapf_texture_plan {
    unsigned result;        /* HapResult: when != 0 nothing else is valid except format */
    unsigned format;        /* HapTextureFormat constant, 0 if unknown */
    unsigned mode;          /* HAPGPU_JOB_* */
    uint64_t section_offset;/* frame offset of the texture section's payload */
    uint32_t section_length;
    /* HAPGPU_JOB_COMPLEX only: */
    int chunk_count;
    uint64_t payload_offset;/* frame offset of the first chunk ("frame_data") */
    uint64_t payload_length;
    HapGpuChunkIn *chunks;  /* malloc'ed, chunk_count entries (unit_* / frag_first not filled) */
    uint64_t frag_table_offset; /* frame offset of the u32 fragment sizes, 0 if absent */
    uint32_t frag_entries;
    uint32_t frag_log2;
    uint32_t frag_gran_log2;/* 1: the table promises 16-bit granular element streams */
    uint32_t frag_window256;/* copy offsets never exceed this many 256-byte units (0: no promise) */
    uint32_t frag_fields;   /* table version 3 ("field streams"): block layout, 4 = [2,6,4,4], 2 = [4,4], 6 = [2,6], 8 = [4,4,4,4]; else 0 */
    uint64_t frag_tiles_offset; /* frame offset of the group tables (96 bytes per fragment entry), 0 if absent */
    unsigned unit_count;    /* filled by the batch layer: GPU work units reserved for this texture */
} hapf_texture_plan;

void hapf_plan_free(hapf_texture_plan *p);

int hapf_read_section(const uint8_t *p, uint32_t available, hapf_section *out);
void hapf_write_section(uint8_t *p, unsigned header_len, uint32_t length, unsigned type);
unsigned hapf_format_from_nibble(unsigned nibble);
unsigned hapf_nibble_from_format(unsigned format);
size_t hapf_snappy_bound(size_t n);
size_t hapf_instructions_length(unsigned chunks);
unsigned hapf_limit_chunk_count(size_t bytes, unsigned format, unsigned chunks);
size_t hapf_texture_bound(size_t bytes, unsigned format, unsigned compressor, unsigned chunks);

/* Locates texture `index` (reference hap.c:932-991). Returns a HapResult. */
unsigned hapf_locate(hapf_reader *r, uint32_t frame_bytes, unsigned index,
                     uint64_t *section_offset, uint32_t *section_length, unsigned *section_type);
/* Plan for texture `index`.  want_chunks != 0: HapDecode semantics (reference
 * hap.c:732-838 minus the payload work; unknown format nibble is Bad_Frame).
 * want_chunks == 0: HapGetFrameTextureChunkCount semantics (reference
 * hap.c:1128-1188; format nibble not examined, chunk list not built). */
void hapf_plan_texture(hapf_reader *r, uint32_t frame_bytes, unsigned index, int want_chunks,
                       hapf_texture_plan *plan);
/* reference hap.c:1042-1087 */
unsigned hapf_texture_count(hapf_reader *r, unsigned long frame_bytes, unsigned *count);

#endif
