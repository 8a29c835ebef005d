/*
 * hap_batch.h -- internal interface between hap_api.c (public symbols) and
 * hap_batch.c (GPU orchestration).  Pure C.
 */
#ifndef HAP_BATCH_H
#define HAP_BATCH_H

#include "../../include/hap_gpu.h"
#include "hap_frame.h"
#include "hapgpu_abi.h"

struct HapGpuContext {
    hapgpu_rt *rt;
    unsigned frag_log2;
    unsigned byte_granular;   /* HAP_AMD_BYTE_GRANULAR=1: never emit 16-bit granular element streams */
    unsigned position_lanes;  /* HAP_AMD_POSITION_LANES: never use the field-per-lane compressor */
    unsigned rgtc1_fields;    /* layout of RGTC1 planes for the block compressor: 26 = [2, 6] (default), 44 = [4, 4], 0 = position lanes (HAP_AMD_RGTC1_LAYOUT) */
    unsigned no_block_scan;   /* HAP_AMD_NO_BLOCK_SCAN: whole-stream units stay whole (A/B runs) */
    unsigned no_fusion;       /* HAP_AMD_NO_FUSION: RGBA calls run the block encoder as a pass of its own (A/B runs) */
    unsigned placing_min_frames; /* HAP_AMD_PLACING_MIN_FRAMES (default 8; 12 until round 5): batches below it gather (with most of a
                                    frame in flight at once its wavefronts wait for each other longer than the gather pass
                                    takes; at 8 frames the two break even for a call by itself, and in a pipelined step --
                                    the next batch's decode kernels fill the waiting wavefronts' slots -- placing wins:
                                    0.545 against 0.559 ms per step of 8 8K frames) */
    unsigned placing_holdoff; /* calls left that gather although they could place: the last placing call encoded most frames twice */
    unsigned placing_holdoff_calls; /* HAP_AMD_PLACING_HOLDOFF (default 8): how many */
    unsigned no_placing;      /* HAP_AMD_NO_PLACING: compressed fragments go to slots and are gathered (A/B runs; also set
                                 while a frame whose chunks did not all shrink is encoded again) */
    unsigned no_half_tiles;   /* HAP_AMD_NO_HALF_TILES: fragment table version 1 even for field streams (A/B runs) */
    /* chunk marks collected from the client's HapDecodeCallback, handed to the retry of a frame whose fragment
       table turned out wrong: the callback is invoked exactly once per HapDecode, as in the reference */
    unsigned long placement_retries; /* frames encoded a second time, through slots (a chunk of theirs was stored raw) */
    unsigned long placement_timeouts; /* ... of which: a wavefront waited for its predecessors' sizes longer than the bound */
    unsigned placing_off;     /* after such a timeout: encode calls left that gather before placing is tried again */
    unsigned long table_fallbacks;   /* frames decoded again without their fragment table (it did not match) */
    const unsigned char *preset_marks;
    unsigned preset_count;
    /* block encode of a batch, handed from hapb_encode_rgba to hapb_encode so that the whole call is one launch
       sequence (recorded and replayed as one HIP graph) */
    const struct HapbBlockEncodeJob *block_encode_job;
    /* texture index of every entry of the next hapb_decode call (NULL: its `index` argument for all) */
    const unsigned *decode_indices;
    /* HapGpuEncodeFramesRGBABegin / HapGpuEncodeFramesFinish: the launched half of an encode call whose results have
       not been asked for yet; while there is one the context takes no other call */
    unsigned defer_encode;
    struct HapbEncodePending *pending_encode;
};

typedef struct HapbBlockEncodeJob {
    const uint64_t *host_table;    /* pinned: [sources][outputs of texture 0][of texture 1], frame_count each */
    uint64_t *device_table;
    unsigned frame_count, count, width, height, formats[2];
    unsigned long row_bytes;
    int wide;
} HapbBlockEncodeJob;

/* What hapb_encode leaves for hapb_encode_complete: the call's arguments (copies: the client's arrays need not outlive
   the first half -- except the two it fills, output_used and results) and the state of its launches. */
typedef struct HapbEncodePending {
    unsigned frame_count, count, flags, live, placed, first_error, launch_rc;
    int inputs_are_device, has_job;
    const void **inputs;
    void **outputs;
    unsigned long *output_bytes;
    unsigned long input_bytes[2];
    unsigned formats[2], compressors[2], chunk_counts[2];
    unsigned long *output_used;       /* the client's */
    unsigned *results;                /* the client's */
    unsigned *live_index;
    size_t *stage_off_out;
    HapGpuFrameEnc *hframes;          /* pinned scratch: stays as it is while the context takes no other call */
    uint8_t *out_stage;
    HapbBlockEncodeJob job;           /* a call that started from pictures (has_job) */
} HapbEncodePending;
unsigned hapb_encode_complete(HapGpuContext *ctx, HapbEncodePending *pending);
/* ... or lets go of it without touching the client's arrays (the context is being destroyed): waits for the launches, frees `pending` */
void hapb_encode_abandon(HapGpuContext *ctx, HapbEncodePending *pending);

/* inputs_are_device != 0: every input pointer is known to be device memory (skips classification) */
unsigned hapb_encode(HapGpuContext *ctx, unsigned frame_count, unsigned count,
                     const void *const *inputs, const unsigned long *input_bytes,
                     const unsigned *formats, const unsigned *compressors, const unsigned *chunk_counts,
                     void *const *outputs, const unsigned long *output_bytes,
                     unsigned long *output_used, unsigned *results, unsigned flags,
                     int inputs_are_device);
unsigned hapb_compress_rgba(HapGpuContext *ctx, const void *rgba, unsigned width, unsigned height,
                            unsigned long row_bytes, unsigned format, void *output,
                            unsigned long output_bytes, unsigned long *used, int synchronise);
unsigned hapb_decompress_rgba(HapGpuContext *ctx, const void *texture, unsigned long texture_bytes, unsigned format,
                              const void *alpha, unsigned long alpha_bytes, unsigned width, unsigned height,
                              void *rgba, unsigned long row_bytes);
unsigned hapb_encode_rgba(HapGpuContext *ctx, unsigned frame_count, const void *const *rgba_frames,
                          unsigned width, unsigned height, unsigned long row_bytes, unsigned count,
                          const unsigned *formats, const unsigned *compressors, const unsigned *chunk_counts,
                          void *const *outputs, const unsigned long *output_bytes,
                          unsigned long *output_used, unsigned *results, unsigned flags);
/* callback/callback_info: only honoured for frame_count == 1 (the hap.h HapDecode path) */
unsigned hapb_decode(HapGpuContext *ctx, unsigned frame_count, const void *const *inputs,
                     const unsigned long *input_bytes, unsigned index, void *const *outputs,
                     const unsigned long *output_bytes, unsigned long *output_used,
                     unsigned *output_formats, unsigned *results, unsigned flags,
                     HapDecodeCallback callback, void *callback_info);

/* frames -> RGBA8 pictures (texture_count 2: Hap Q Alpha frames, colour + RGTC1 alpha plane) */
unsigned hapb_decode_rgba(HapGpuContext *ctx, unsigned frame_count, const void *const *inputs,
                          const unsigned long *input_bytes, unsigned texture_count, void *const *rgba_frames,
                          unsigned width, unsigned height, unsigned long row_bytes, unsigned *results, unsigned flags);

/* groups and output in device memory: tables through the host, payloads device to device */
unsigned hapb_join_device(HapGpuContext *ctx, unsigned group_count, const void *const *frames,
                          const unsigned long *frame_bytes, void *output, unsigned long output_bytes,
                          unsigned long *output_used);

/* hap_join.c: the join proper, reading the groups' headers through `readers` (host views, or device frames with a
 * fetch callback) and writing through a sink: `put` = bytes made here to output offset, `move` = bytes of group g's
 * frame to output offset.  Non-zero from a sink callback ends the join with Internal_Error. */
typedef struct hapj_sink {
    void *user;
    int (*put)(void *user, uint64_t dst_off, const void *src, size_t len);
    int (*move)(void *user, unsigned group, uint64_t src_off, uint64_t dst_off, size_t len);
} hapj_sink;
unsigned hapj_join(unsigned groupCount, hapf_reader *readers, const unsigned long *groupFramesBytes,
                   const hapj_sink *sink, unsigned long outputBufferBytes, unsigned long *outputBufferBytesUsed);

#endif
