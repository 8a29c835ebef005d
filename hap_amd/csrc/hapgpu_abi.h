/*
 * hapgpu_abi.h -- the thin C ABI between the host-side C code (hap_api.c,
 * hap_batch.c, hap_frame.c: pure C99, no HIP headers) and the HIP translation
 * unit (hapgpu_runtime.hip + kernels).  Plain pointers, integers and PODs
 * only.  Structures marked [device] are laid out identically for gcc/clang
 * host code and hipcc device code (fixed-width members, natural alignment).
 */
#ifndef HAPGPU_ABI_H
#define HAPGPU_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Environment switches that exist for A/B measurements only (copy kernels against hipMemcpyAsync, ring sizes, compressor
   layouts ...) are read by measurement builds (tools/build_variants.sh, -DHAP_MEASUREMENT_BUILD); the product library does
   not look at them: every switch it does read is listed in INTEGRATION.md and exercised by a test
   (tests/test_cabi_cpu.py::test_every_environment_switch_is_documented_and_tested). */
#ifdef HAP_MEASUREMENT_BUILD
#define HAP_AB_ENV(name) getenv(name)
#else
#define HAP_AB_ENV(name) ((const char *)0)
#endif

/* ---- container constants (reference hap.c:41-51, 84-88) ---- */
#define HAP_NIBBLE_NONE 0xAu
#define HAP_NIBBLE_SNAPPY 0xBu
#define HAP_NIBBLE_COMPLEX 0xCu
#define HAP_SECTION_MULTI 0x0Du
#define HAP_SECTION_INSTRUCTIONS 0x01u
#define HAP_SECTION_COMPRESSORS 0x02u
#define HAP_SECTION_SIZES 0x03u
#define HAP_SECTION_OFFSETS 0x04u
#define HAP_SECTION_FRAGMENTS 0x46u   /* private: hap_gpu.h */
#define HAP_FRAGMENT_TABLE_VERSION 1u       /* fragment sizes only */
#define HAP_FRAGMENT_TABLE_VERSION_FIELDS 4u /* + a group table per fragment: "field streams".  (Version 2, one byte per
                                                half-tile, and version 3, 96-byte group tables without the groups' output
                                                bytes, were written by earlier builds: version 2 tables are ignored, of
                                                version 3 the fragment sizes are used.) */
#define HAP_FRAGMENT_TABLE_VERSION_FIELDS_R4 3u
#define HAP_GROUP_TABLE_BYTES_R4 96u
#define HAP_GROUP_TABLE_BYTES 196u           /* 64 groups of equally many elements, 24 bits each -- compressed bytes |
                                                output bytes << 12 -- then LE16 element count, LE16 zero */
#define HAP_HALF_TILE_BYTES 128u
#define HAP_HALF_TILES_PER_FRAGMENT 64u      /* 8 KiB fragments */

/* A fragment's slot in the compressor's scratch: at most 130 bytes per half-tile of elements, then a few bytes per
   lane that stores with nothing to write are pointed at (snappy_compress_blocks.hip) */
#define HAPGPU_SLOT_DATA_BYTES (HAP_HALF_TILES_PER_FRAGMENT * 130u)
#define HAPGPU_SLOT_SCRATCH_BYTES 272u

/* internal status codes beyond HapResult (never returned to API callers) */
#define HAPGPU_STATUS_INDEX_MISMATCH 100u /* fragment index inconsistent: redo without it */

/* ------------------------------------------------------------------ */
/* encode side                                                          */
/* ------------------------------------------------------------------ */

/* [device] one texture of one frame */
typedef struct HapGpuTexEnc {
    uint64_t src;            /* device address of the block-compressed texture */
    uint32_t bytes;          /* texture bytes */
    uint32_t format_nibble;  /* low nibble of the section type (hap.c:45-51) */
    uint32_t compressor;     /* 0 none, 1 snappy (hap.h:50-53) */
    uint32_t chunk_count;    /* already limited (hap.c:277-300) */
    uint32_t chunk_bytes;    /* bytes / chunk_count */
    uint32_t header_len;     /* 4 or 8 (hap.c:398-405, 425-428) */
    uint32_t frags_per_chunk;
    uint32_t frag_first;     /* global index of this texture's first fragment */
    uint32_t emit_index;     /* write the fragment-size section */
    uint32_t reserved;       /* bit 20: "field stream": no element crosses a 128-byte half-tile and the compressor
                                writes every fragment's group table (fragment table version 4);
                                bits 16..19: fields per block for the field-per-lane compressor (0: position per lane,
                                2: RGTC1 layout, 4: DXT5 / YCoCg-DXT5, 10: DXT1, 12: opaque 16-byte blocks); bits 8..15: match window in 256-byte units (0 = whole fragment); bits 0..7:
                                granularity_log2 of the element stream: 0 = bytes, 1 = every position, offset
                                and length even (lets the decoder move 16 bits per lane);
                                bits 24..26: 0 = the texture is at src; else the block-per-lane compressor makes it from
                                the frame's RGBA picture (1 DXT1, 2 DXT5, 3 scaled YCoCg-DXT5, 4 RGTC1 from alpha) and
                                writes it to src on the way;
                                bit 27: "placed": the block-per-lane compressor writes every fragment's stream at its final
                                place in the frame (the sizes of everything before it: frag_sizes entries with
                                HAPGPU_FRAG_PUBLISHED set, chunk_acc words of the chunks before) instead of a slot, on the
                                assumption that every chunk shrinks; the pack kernel reports HAPGPU_STATUS_NOT_PLACED
                                for a frame where one did not (or a wavefront gave up waiting), and the host encodes that
                                frame again through slots.  First texture of a frame only. */
} HapGpuTexEnc;
#define HAPGPU_FRAG_PUBLISHED 0x80000000u
#define HAPGPU_STATUS_NOT_PLACED 0x7E50u

/* [device] one frame */
typedef struct HapGpuFrameEnc {
    uint64_t dst;            /* device address of the frame buffer */
    uint64_t dst_cap;
    uint32_t tex_count;      /* 1 or 2 */
    uint32_t outer_header_len; /* 0 (single texture), 4 or 8 (hap.c:562-576) */
    HapGpuTexEnc tex[2];
    /* results */
    uint64_t bytes_used;
    uint32_t status;
    uint32_t reserved;
    /* textures whose reserved bits 24..26 are set are made from this RGBA8 picture by the second stage itself */
    uint64_t rgba;
    uint32_t rgba_row_bytes; /* (the picture is smaller than 4 GiB) */
    uint32_t rgba_blocks_x;  /* width / 4 */
    /* textures whose reserved bit 27 is set: one 64-bit word per chunk of the frame (zero before the launch), in which
       the compressor's wavefronts add up fragments << 32 | bytes as they learn their sizes */
    uint64_t chunk_acc;
} HapGpuFrameEnc;

/* [device] one byte-range move of the gather pass (one per fragment) */
typedef struct HapGpuCopyEntry {
    uint64_t src;
    uint64_t dst;
    uint32_t len;
    uint32_t reserved;
} HapGpuCopyEntry;

/* ------------------------------------------------------------------ */
/* decode side                                                          */
/* ------------------------------------------------------------------ */

/* [device] one chunk as listed by the frame's tables (hap.c:794-809) */
typedef struct HapGpuChunkIn {
    uint32_t src_off;        /* from the start of the payload ("frame_data") */
    uint32_t src_len;
    uint32_t codec;          /* compressor table byte; bit 31 set = not requested by the client */
    uint32_t unit_first;     /* first unit slot of this chunk (relative to the job) */
    uint32_t unit_count;     /* slots reserved for it */
    uint32_t frag_first;     /* first fragment-table entry of this chunk, if the frame has one */
    /* written by the plan kernel for the expansion kernel (one wavefront per chunk turns the chunk's
       fragment-table entries into units) */
    uint32_t plan_expand;    /* 1: expand with the fragment table */
    uint32_t plan_hdr;       /* bytes of the chunk's varint length prefix */
    uint32_t plan_out_len;   /* decoded bytes of the chunk */
    uint32_t reserved;
    uint64_t plan_out_off;   /* where they go, relative to the job's dst */
} HapGpuChunkIn;

#define HAPGPU_JOB_COMPLEX 0u
#define HAPGPU_JOB_SNAPPY 1u  /* 0xB_: one Snappy stream (hap.c:885-904) */
#define HAPGPU_JOB_RAW 2u     /* 0xA_ (hap.c:905-916) */

/* [device] one texture to decode */
typedef struct HapGpuDecodeJob {
    uint64_t payload;        /* device address of the first chunk's bytes */
    uint64_t payload_len;    /* bytes from payload to the end of the texture section */
    uint64_t dst;
    uint64_t dst_cap;
    uint64_t chunks;         /* device address of HapGpuChunkIn[chunk_count] */
    uint64_t frag_sizes;     /* device address of the u32 fragment-size entries inside the frame, or 0 */
    uint64_t units;          /* device address of this job's HapGpuDecodeUnit[unit_count] */
    uint32_t chunk_count;
    uint32_t mode;           /* HAPGPU_JOB_* */
    uint32_t frag_log2;
    uint32_t frag_entries;   /* number of fragment-size entries */
    uint32_t unit_count;
    uint32_t reserved;       /* bits 0..7: granularity_log2 announced by the fragment table (0 bytes, 1 16-bit,
                                2 32-bit); bits 8..15: its match window in 256-byte units (0 = none announced);
                                bit 16: no table, but every chunk is as short as a fragment: group_tables is scratch
                                (HAP_GROUP_TABLE_BYTES per unit of the CALL) that hapgpu_k_guess_group_tables fills,
                                fields_period the layout the texture format implies */
    /* results */
    uint64_t bytes_used;
    uint32_t status;         /* HapResult or HAPGPU_STATUS_* */
    uint32_t fields_period;  /* 4 / 2 / 6 / 8: the table is version 3 and promises [2,6,4,4] / [4,4] / [2,6] / [4,4,4,4] field streams; 0 otherwise */
    uint64_t group_tables;     /* device address of the group tables inside the frame (96 bytes per fragment entry), or 0 */
} HapGpuDecodeJob;

#define HAPGPU_UNIT_SKIP 0u
#define HAPGPU_UNIT_SNAPPY_STREAM 1u   /* varint header + elements */
#define HAPGPU_UNIT_SNAPPY_FRAGMENT 2u /* bare elements producing exactly dst_len bytes */
#define HAPGPU_UNIT_COPY 3u
#define HAPGPU_UNIT_SNAPPY_FRAGMENT16 4u /* fragment whose elements are all 16-bit granular */
#define HAPGPU_UNIT_SNAPPY_FRAGMENT32 5u /* ... all 32-bit granular */
#define HAPGPU_UNIT_SNAPPY_FIELDS4 6u   /* fragment of a field stream, 16-byte blocks of 2 + 6 + 4 + 4 bytes; aux = its group table */
#define HAPGPU_UNIT_SNAPPY_FIELDS2 7u   /* ... 8-byte blocks of 4 + 4 bytes */
#define HAPGPU_UNIT_SNAPPY_FIELDS26 8u  /* ... 8-byte blocks of 2 + 6 bytes */
#define HAPGPU_UNIT_SNAPPY_FIELDS44 10u /* ... 16-byte blocks of 4 + 4 + 4 + 4 bytes (opaque formats) */
#define HAPGPU_UNIT_SNAPPY_BLOCK 9u     /* one 64 KiB block of another encoder's stream (or one 8 KiB block of a table-less
                                           stream of this library), found by the block scan: bare elements, copies stay
                                           inside the block; decoded by the whole-stream kernel.  aux = its
                                           HapGpuScanChunk, reserved = block number (| HAPGPU_BLOCK_FINE), src = the stream */
#define HAPGPU_BLOCK_FINE (1ull << 32)  /* flag in reserved: the unit is an 8 KiB block */
#define HAPGPU_SCAN_FINE 8192u          /* the block scan's fine granularity (one fragment of this library's streams) */
#define HAPGPU_UNIT_WINDOWED 0x10u       /* flag on the three fragment kinds: every copy offset is <= 3 KiB, so an 8 KiB
                                            fragment decodes through a 4 KiB LDS ring (twice the waves per CU) */
#define HAP_FRAGMENT_WINDOW_256 12u      /* that window in 256-byte units, as written to the fragment table */

/* Block scan of another encoder's Snappy stream (snappy_decode.hip): the stream's compressed bytes are looked at in
   segments of HAPGPU_SCAN_SEGMENT bytes (of the 16-byte aligned address range that holds them). */
#ifndef HAPGPU_SCAN_SEGMENT
#define HAPGPU_SCAN_SEGMENT 4096u
#endif
typedef struct HapGpuScanChunk {
    /* filled by the host */
    uint32_t unit;           /* index of the stream's whole-stream unit in the call's unit array */
    uint32_t seg_first;      /* first of its seg_count segment records */
    uint32_t seg_count;      /* >= segments the stream can touch: (src_len + 15 + 4095) / 4096 */
    uint32_t slots;          /* unit slots for 64 KiB blocks, directly behind the stream unit */
    uint64_t bpos;           /* device address of fine_slots + 1 words: compressed position where each 8 KiB of output
                                begins (a 64 KiB block begins at every eighth) */
    /* written by the device (the host sends zeros) */
    uint32_t ok;             /* the element chain was followed to the stream's end and the lengths agree */
    uint32_t expected;       /* 64 KiB blocks */
    uint32_t found;          /* 64 KiB block starts that fall on an element boundary: all of them = BLOCK units run */
    /* host */
    uint32_t fine_slots;     /* unit slots for 8 KiB blocks, at fine_unit_first of the call's unit array (behind all the
                                ordinary units): streams written by this library without a fragment table have an
                                element boundary at every 8 KiB (its fragments) -- eight times the units of
                                libsnappy's 64 KiB blocks */
    /* device */
    uint32_t expected_fine;  /* 8 KiB blocks */
    uint32_t found_fine;     /* 8 KiB marks on element boundaries: all of them = the fine BLOCK units run first */
    uint32_t fine_failed;    /* ... and one of them met a copy that reaches before its block (marks on element boundaries
                                do not make a stream's 8 KiB pieces independent): the 64 KiB blocks / the stream unit
                                of the launch's second phase decode the stream instead */
    uint32_t fine_unit_first; /* device: index of the stream's first fine unit slot in the call's unit array (from the pool) */
    uint32_t seg_bytes;      /* host: the call's segment size, HAPGPU_SCAN_SEGMENT or half of it (0: HAPGPU_SCAN_SEGMENT) */
    uint32_t probe_found;    /* of the first two 8 KiB marks (output positions 8192 and 16384): the rest of the fine
                                marks are only looked for when both fall on element boundaries -- in a libsnappy
                                stream they hardly ever do, and its scan then costs what it did with 64 KiB marks only */
} HapGpuScanChunk;

typedef struct HapGpuScanSegment {   /* device only */
    uint32_t exit_coord;     /* where the segment's (guessed) chain left the segment */
    uint32_t cum_total;      /* output bytes that chain produced from its start to there */
    uint32_t flags;          /* 1: the chain met something that is not an element; 2: it reached the end of the input */
    uint32_t elements;       /* elements of that chain from its start to there */
} HapGpuScanSegment;
/* (per segment, in an array of its own, 4 x uint32: the window in which the true chain joined the recorded one --
   0xFFFFFFFF: never --, the absolute output position and the absolute element number of the recorded chain's zero, and
   1 where the true chain may have entered the segment in front of that window.  A window's record, 64 bits: byte of the
   window at which the chain entered it | output bytes of the chain so far << 8 | its elements so far << 40) */

/* [device] one wavefront's worth of decode work */
typedef struct HapGpuDecodeUnit {
    uint64_t src;
    uint64_t dst;
    uint32_t src_len;
    uint32_t dst_len;
    uint32_t kind;           /* HAPGPU_UNIT_* */
    uint32_t job;            /* index of the owning job (status word) */
    uint64_t aux;            /* FIELDS units: device address of the fragment's group table (HAP_GROUP_TABLE_BYTES);
                                STREAM units: number of SKIP slots that follow for the block scan's BLOCK units */
    /* reserved: fragment units: readable bytes after the fragment (<= 15); STREAM units: 0 or the HapGpuScanChunk
       that decides whether the stream unit or its BLOCK units run */
    uint64_t reserved;
} HapGpuDecodeUnit;

/* ------------------------------------------------------------------ */
/* runtime + launchers (implemented in hapgpu_runtime.hip)              */
/* ------------------------------------------------------------------ */
typedef struct hapgpu_rt hapgpu_rt;

int hapgpu_rt_create(int device, hapgpu_rt **rt);
void hapgpu_rt_destroy(hapgpu_rt *rt);
/* 1: HIP device memory (usable by kernels in place), 0: host memory */
int hapgpu_rt_is_device_ptr(hapgpu_rt *rt, const void *p);
/* grow-only scratch arenas, slot 0..15; contents undefined after growth */
void *hapgpu_rt_device_scratch(hapgpu_rt *rt, int slot, size_t bytes);
void *hapgpu_rt_pinned_scratch(hapgpu_rt *rt, int slot, size_t bytes);
int hapgpu_rt_h2d(hapgpu_rt *rt, void *dst, const void *src, size_t bytes);
int hapgpu_rt_d2h(hapgpu_rt *rt, void *dst, const void *src, size_t bytes);
int hapgpu_rt_d2h_rows(hapgpu_rt *rt, void *dst, size_t dpitch, const void *src, size_t spitch, size_t row, size_t rows);
int hapgpu_rt_d2d(hapgpu_rt *rt, void *dst, const void *src, size_t bytes);
int hapgpu_rt_zero(hapgpu_rt *rt, void *dst, size_t bytes);
/* 1: kernels may be handed pinned-scratch addresses directly (no copy in front of or behind them) */
int hapgpu_rt_pinned_is_mapped(hapgpu_rt *rt);
int hapgpu_rt_sync(hapgpu_rt *rt);
/* 64 KiB blocks of other encoders' streams decoded by a workgroup each since the runtime was made (waits for the stream) */
unsigned hapgpu_rt_resolved_blocks(hapgpu_rt *rt);
void hapgpu_rt_lock(hapgpu_rt *rt);
void hapgpu_rt_unlock(hapgpu_rt *rt);
/* 0: the lock was free and is now held by the caller */
int hapgpu_rt_trylock(hapgpu_rt *rt);
int hapgpu_rt_device(hapgpu_rt *rt);
/* Launch sequences recorded as HIP graphs (hapgpu_runtime.hip).  key: everything the sequence's kernel arguments and
 * copy sizes depend on.  _begin: 1 = a recorded sequence was launched (skip the launches), 0 = recording (issue the
 * launches, then _end), 2 = launch as usual.  _end(failed): 0 = instantiated, remembered and launched. */
int hapgpu_rt_graph_begin(hapgpu_rt *rt, uint64_t key);
int hapgpu_rt_graph_end(hapgpu_rt *rt, uint64_t key, int failed);
/* no more recordings for this runtime (its recorded sequences stay usable): after a capture / instantiate / launch failure */
void hapgpu_rt_graphs_disable(hapgpu_rt *rt);

/* kernels: all asynchronous on the runtime's stream; 0 = launched */
int hapgpu_k_block_encode(hapgpu_rt *rt, const void *rgba, unsigned width, unsigned height,
                          size_t row_bytes, unsigned hap_texture_format, void *out);
/* pictures of one geometry in one launch; sources / outputs: DEVICE arrays of device addresses (0 = skip) */
int hapgpu_k_block_encode_batch(hapgpu_rt *rt, const uint64_t *sources, const uint64_t *outputs, unsigned pictures,
                                unsigned width, unsigned height, size_t row_bytes, unsigned hap_texture_format, int wide);
/* Hap Q Alpha: scaled YCoCg-DXT5 + RGTC1 alpha plane of every picture from one read of its RGBA */
int hapgpu_k_block_encode_batch_ycocg_alpha(hapgpu_rt *rt, const uint64_t *sources, const uint64_t *colour_outputs,
                                            const uint64_t *alpha_outputs, unsigned pictures, unsigned width,
                                            unsigned height, size_t row_bytes, int wide);
int hapgpu_k_block_decode(hapgpu_rt *rt, const void *blocks, const void *alpha, unsigned width, unsigned height,
                          unsigned hap_texture_format, void *rgba, size_t row_bytes);
/* pictures of one format and geometry in one launch; table: DEVICE array of device addresses, [textures][alpha planes]
   [pictures], `pictures` entries each (texture 0 = skip the picture) */
int hapgpu_k_block_decode_batch(hapgpu_rt *rt, const uint64_t *table, unsigned pictures, int with_alpha, unsigned width,
                                unsigned height, unsigned hap_texture_format, size_t row_bytes);
/* group_tables: HAP_GROUP_TABLE_BYTES bytes per fragment (same indexing as frag_sizes), written for textures whose reserved bit 20 is set */
int hapgpu_k_snappy_compress(hapgpu_rt *rt, const HapGpuFrameEnc *frames, unsigned frame_count,
                             unsigned max_frags_per_texture, unsigned frag_log2,
                             void *slots, unsigned slot_stride, uint32_t *frag_sizes, uint8_t *group_tables,
                             unsigned granularity_mask /* bit g (0..2) set: some position-per-lane texture has granularity_log2 == g; bit 4 / 5 / 6 / 7: some texture uses the field-per-lane kernel ([2,6] / [2,6,4,4] / [4,4] / [4,4,4,4] fields per block); bits 8..: textures per frame */);
/* copies: one entry per fragment, then (from index extra_first) chunks_per_frame entries per frame for the
 * group tables of field streams */
int hapgpu_k_frame_pack(hapgpu_rt *rt, HapGpuFrameEnc *frames, unsigned frame_count,
                        unsigned frag_log2, const void *slots, unsigned slot_stride,
                        const uint32_t *frag_sizes, const uint8_t *group_tables, HapGpuCopyEntry *copies,
                        unsigned extra_first, unsigned chunks_per_frame, unsigned max_chunks_per_texture,
                        unsigned textures, void *pack_scratch);
/* bytes of pack_scratch needed per chunk (frame_count * chunks_per_frame of them) */
unsigned hapgpu_pack_scratch_bytes_per_chunk(void);
int hapgpu_k_frame_gather(hapgpu_rt *rt, const HapGpuCopyEntry *copies, unsigned count);
/* first `prefix` bytes of every device-resident frame (0 pointer = skip) -> out_dev + i * prefix */
int hapgpu_k_gather_prefixes(hapgpu_rt *rt, const uint64_t *frames_dev, const uint64_t *lengths_dev,
                             unsigned count, unsigned prefix, void *out_dev);
/* ... and, for entries with far_dev[f] != 0, the bytes at the second texture's section when it starts beyond the first
   prefix (far_at_dev[f] = its offset, 0 = none / inside the first prefix) */
int hapgpu_k_gather_prefixes_far(hapgpu_rt *rt, const uint64_t *frames_dev, const uint64_t *lengths_dev,
                                 unsigned count, unsigned prefix, void *out_dev, const uint8_t *far_dev,
                                 void *out2_dev, uint64_t *far_at_dev);
/* clears `units` (all SKIP) then plans every job; max_chunks: largest chunk_count among the jobs */
int hapgpu_k_decode_plan(hapgpu_rt *rt, HapGpuDecodeJob *jobs, unsigned job_count,
                         HapGpuDecodeUnit *units, unsigned unit_count, unsigned max_chunks);
/* splits whole-stream units that consist of independent 64 KiB blocks (what libsnappy writes) into BLOCK units,
 * using the slots reserved behind them; streams that do not qualify stay as they are.  chunks: one entry per stream
 * (host-filled part copied to the device by the caller); segs / recs / joins: device scratch of seg_total entries /
 * seg_total * 64 words of 8 bytes / seg_total * 8 bytes */
int hapgpu_k_scan_blocks(hapgpu_rt *rt, HapGpuDecodeUnit *units, const HapGpuDecodeJob *jobs, HapGpuScanChunk *chunks,
                         unsigned chunk_count, HapGpuScanSegment *segs, void *recs, void *joins, unsigned seg_total,
                         uint32_t *fine_work /* [0]: count (zero on entry), then the unit indices of the 8 KiB blocks to decode
                                                (room for fine_pool of them), then the pool's cursor (zero on entry) */,
                         unsigned fine_first, unsigned fine_pool /* the 8 KiB blocks' unit slots: units[fine_first .. + fine_pool),
                                                                    handed out to the streams on the device */);
/* frag_log2: fragment size of the batch's FRAGMENT units (0: none present);
 * fragment_kinds: bit g set = fragments of granularity_log2 g present */
/* fragment_kinds bits 8 / 9 / 10: field-stream units of [2,6,4,4] / [4,4] / [2,6] blocks present */
/* any_stream_or_copy_units: 0 none, 1 whole streams / raw copies, 2 the same with block-scanned streams among them */
int hapgpu_k_snappy_decode(hapgpu_rt *rt, const HapGpuDecodeUnit *units, unsigned unit_count,
                           HapGpuDecodeJob *jobs, unsigned frag_log2, unsigned fragment_kinds,
                           int any_stream_or_copy_units,
                           const uint32_t *fine_work, unsigned fine_slots /* the block scan's list and its capacity (0: none) */);

/* measurement */
void hapgpu_rt_set_profiling(hapgpu_rt *rt, int enable);
int hapgpu_rt_collect_profile(hapgpu_rt *rt, unsigned long *launches, double *ms, unsigned classes);
/* group tables for the STREAM units of jobs whose reserved bit 16 is set (fragments that came as chunks of their own,
   without a private table): units that turn out to be field streams become FIELDS units (snappy_decode_fields.hip) */
/* (work != NULL: only the units the block scan listed there -- the 8 KiB pieces of table-less streams of this library;
   unit_count then spans the fine region behind the ordinary units as well) */
int hapgpu_k_guess_group_tables(hapgpu_rt *rt, HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                const uint32_t *work, unsigned work_slots);
int hapgpu_rt_timer_start(hapgpu_rt *rt);
int hapgpu_rt_timer_stop(hapgpu_rt *rt, double *ms);

#ifdef __cplusplus
}
#endif
#endif
