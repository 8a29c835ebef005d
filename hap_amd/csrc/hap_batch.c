/*
 * hap_batch.c -- host-side orchestration of the GPU hot path (pure C99).
 *
 * Implements the batched entry points of include/hap_gpu.h; the six hap.h
 * functions in hap_api.c are thin wrappers around the batch-of-one case.
 * The host does only container arithmetic (header lengths, chunk-count
 * limiting, section location, table parsing -- see hap_frame.c) and fills the
 * descriptor arrays of hapgpu_abi.h; every byte of texture payload is moved,
 * compressed or decompressed by HIP kernels.  There is no CPU fallback: when
 * no HIP device is present the functions fail with HapResult_Internal_Error.
 *
 * Reference behaviour mirrored here: argument checks and result codes of
 * HapEncode (hap.c:506-604, 355-504) and HapDecode (hap.c:993-1040, 732-930).
 */
#include "hap_batch.h"
#include "measurement_guard.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
    D_FRAMES = 0, D_SLOTS, D_FRAGSIZES, D_COPIES, D_TEX_STAGE, D_FRAME_STAGE, D_BC_TEX, D_RGBA_STAGE,
    D_JOBS, D_CHUNKS, D_UNITS, D_IN_STAGE, D_OUT_STAGE, D_PTRS, D_PREFIX, D_BC_PTRS
};
#define D_GROUPTABLES D_PTRS   /* encode-only arenas in decode-only slots: one call never needs both */
#define D_PACK D_PREFIX
#define D_CHUNK_ACC D_JOBS
#define D_SCAN D_SLOTS       /* ... and the decoder's block-scan arena in an encode-only one */
#define D_GUESS D_FRAGSIZES  /* ... and the group tables the decoder makes for fragments that came without */
#define P_SCAN P_FRAMES
enum { P_FRAMES = 0, P_JOBS, P_CHUNKS, P_PREFIX, P_PTRS, P_BC_PTRS, P_PREFIX2 };   /* (8, 9: hap_sequence.c) */

#ifdef HAPB_TRACE
#include <stdio.h>
#include <time.h>
static double hapb_now_us(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
#define HAPB_MARK(name) do { double t_ = hapb_now_us(); fprintf(stderr, "  [decode %u] %-10s +%.1f us\n", frame_count, name, t_ - hapb_t0); hapb_t0 = t_; } while (0)
#else
#define HAPB_MARK(name) do { } while (0)
#endif
#define PREFIX_BYTES 2048u   /* headers + tables of a frame with up to ~400 chunks; larger ones are fetched on demand */
#define PREFIX_BATCH_MAX_BYTES ((size_t)32u << 20)   /* the longer second prefix, all frames of a call together (x2: a far window each) */
#define COPY_PIECE 65536u

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int is_dev(HapGpuContext *c, const void *p) { return hapgpu_rt_is_device_ptr(c->rt, p); }

/* A context whose last encode call has been begun and not finished (HapGpuEncodeFramesRGBABegin) holds that call's
   tables and scratch: it takes no other call until HapGpuEncodeFramesFinish. */
static int context_busy(HapGpuContext *c, unsigned *results, unsigned frame_count)
{
    unsigned f;
    if (!c->pending_encode)
        return 0;
    fprintf(stderr, "hap_amd: the context has an encode call in flight (HapGpuEncodeFramesFinish first)\n");
    for (f = 0; results && f < frame_count; f++)
        results[f] = HapResult_Internal_Error;
    return 1;
}

/* ================================================================== encode */

typedef struct tex_geom {
    unsigned format, compressor, chunk_count, chunk_bytes, header_len, fpc, nibble, gran_log2, field_period, half_tiles;
    unsigned long bytes;
    size_t bound;        /* hap_max_encoded_length for the requested compressor (hap.c:386) */
} tex_geom;

unsigned hapb_encode(HapGpuContext *ctx, unsigned frame_count, unsigned count,
                     const void *const *inputs, const unsigned long *input_bytes,
                     const unsigned *formats, const unsigned *compressors, const unsigned *chunk_counts,
                     void *const *outputs, const unsigned long *output_bytes,
                     unsigned long *output_used, unsigned *results, unsigned flags,
                     int inputs_are_device)
{
    tex_geom g[2];
    unsigned fine_counts[2] = {0u, 0u};
    unsigned i, f, first_error = HapResult_No_Error;
    unsigned outer_header = 0, frags_per_frame = 0, max_frags_per_tex = 0, live = 0, chunks_per_frame = 0;
    int any_half_tiles = 0;
    uint8_t *dgrouptables = NULL;
    void *dpack = NULL;
    unsigned max_chunks_per_tex = 0;
    /* HAPGPU_ENCODE_SMALLER_FILES: 64 KiB fragments (a match may lie 64 KiB back, like libsnappy's), elements on
       16-bit positions anywhere instead of whole block fields, and no fragment table */
    const int smaller = (flags & HAPGPU_ENCODE_SMALLER_FILES) != 0;
    unsigned frag_log2 = smaller ? 16u : ctx->frag_log2, frag_bytes = 1u << frag_log2;
    unsigned slot_stride;
    int any_snappy = 0;
    unsigned gran_mask = 0;
    size_t placed_extent = 0;
    unsigned placed = 0u;                            /* texture 0's fragments are written where they belong in the frame (bit 27) */
    uint64_t *dacc = NULL;
    unsigned fused[2] = {0u, 0u}, fused_mask = 0u;   /* textures the second stage makes from the RGBA itself (code: reserved bits 24..26) */
    size_t stage_in_bytes = 0, stage_out_bytes = 0, frame_raw_bound = 0;
    hapgpu_rt *rt = ctx->rt;
    HapGpuFrameEnc *hframes, *dframes;
    uint8_t *dslots, *tex_stage = NULL, *out_stage = NULL;
    uint32_t *dfragsizes;
    HapGpuCopyEntry *dcopies;
    size_t *stage_off_in = NULL, *stage_off_out = NULL;
    unsigned *live_index = NULL;

    if (frame_count == 0)
        return HapResult_No_Error;
    if (!results)
        return HapResult_Bad_Arguments;
    if (context_busy(ctx, results, frame_count))
        return HapResult_Internal_Error;
    /* frame-independent argument checks, reference hap.c:518-559 */
    {
        unsigned rc = HapResult_No_Error;
        if (count == 0 || count > 2 || !inputs || !input_bytes || !formats || !compressors || !chunk_counts ||
            !outputs || !output_bytes || !output_used)
            rc = HapResult_Bad_Arguments;
        /* (with HAPGPU_ENCODE_FINE_CHUNKS the call's chunk counts are replaced below: whatever they say is not looked at) */
        for (i = 0; rc == HapResult_No_Error && i < count; i++)
            if (chunk_counts[i] == 0 && !((flags & HAPGPU_ENCODE_FINE_CHUNKS) && !smaller && compressors[i] == HapCompressorSnappy))
                rc = HapResult_Bad_Arguments;
        if (rc == HapResult_No_Error && count == 2 &&
            formats[0] != HapTextureFormat_YCoCg_DXT5 && formats[1] != HapTextureFormat_YCoCg_DXT5 &&
            formats[0] != HapTextureFormat_A_RGTC1 && formats[1] != HapTextureFormat_A_RGTC1)
            rc = HapResult_Bad_Arguments;
        /* per-texture checks that do not depend on the frame, reference hap.c:367-385 */
        for (i = 0; rc == HapResult_No_Error && i < count; i++)
            if (input_bytes[i] == 0 || input_bytes[i] > 0xFFFFFFFFul || hapf_nibble_from_format(formats[i]) == 0 ||
                (compressors[i] != HapCompressorNone && compressors[i] != HapCompressorSnappy))
                rc = HapResult_Bad_Arguments;
        if (rc != HapResult_No_Error) {
            for (f = 0; f < frame_count; f++)
                results[f] = rc;
            return rc;
        }
    }
    if (smaller)
        flags &= ~HAPGPU_ENCODE_FRAGMENT_INDEX;
    /* HAPGPU_ENCODE_FINE_CHUNKS: one chunk per Snappy fragment (8 KiB of texture) -- the chunk size table every Hap parser
       reads (hap.c:265-300) then says where each independently compressed piece begins */
    if ((flags & HAPGPU_ENCODE_FINE_CHUNKS) && !smaller) {
        for (i = 0; i < count; i++)
            fine_counts[i] = compressors[i] == HapCompressorSnappy ? HapGpuFineChunkCount(input_bytes[i], formats[i]) : chunk_counts[i];
        chunk_counts = fine_counts;
    }
    /* geometry shared by every frame of the batch */
    if (count == 2) {
        size_t worst = 0;                                             /* hap.c:563-576 */
        for (i = 0; i < count; i++)
            worst += input_bytes[i] + hapf_instructions_length(chunk_counts[i]) + 4u;
        outer_header = worst > 0xFFFFFFu ? 8u : 4u;
    }
    for (i = 0; i < count; i++) {
        tex_geom *t = &g[i];
        t->format = formats[i];
        t->nibble = hapf_nibble_from_format(formats[i]);
        t->compressor = compressors[i];
        t->bytes = input_bytes[i];
        t->bound = hapf_texture_bound(t->bytes, t->format, t->compressor, chunk_counts[i]);
        t->header_len = t->bytes > 0xFFFFFFu ? 8u : 4u;              /* hap.c:398-405 */
        if (t->compressor == HapCompressorSnappy) {
            t->chunk_count = hapf_limit_chunk_count(t->bytes, t->format, chunk_counts[i]);
            t->chunk_bytes = (unsigned)(t->bytes / t->chunk_count);
        }
        if (t->compressor == HapCompressorSnappy && t->chunk_bytes == 0) {
            /* less than one block per chunk (not a real texture): the reference compresses
               zero-length chunks, finds no gain and stores the section as-is (hap.c:478-495) */
            t->compressor = HapCompressorNone;
        }
        if (t->compressor == HapCompressorSnappy) {
            any_snappy = 1;
        } else {
            t->chunk_count = 1;
            t->chunk_bytes = (unsigned)t->bytes;
        }
        t->fpc = (t->chunk_bytes + frag_bytes - 1) / frag_bytes;
        /* 16-bit granular element streams need even chunk sizes (always true for block textures) */
        t->gran_log2 = 0u;
        if (t->compressor == HapCompressorSnappy && !ctx->byte_granular) {
            /* DXT1 blocks are two 4-byte fields (endpoints, indices): everything repeats on 4-byte
               boundaries.  The alpha-style blocks of RGTC1 / DXT5 / YCoCg start their index bytes at
               byte 2, so those streams are 2-byte granular. */
            if ((t->format == HapTextureFormat_RGB_DXT1 || (flags & HAPGPU_ENCODE_COARSE_MATCHES) ||
                 (t->format == HapTextureFormat_A_RGTC1 && ctx->rgtc1_fields && ctx->rgtc1_fields != 26u)) && (t->chunk_bytes & 3u) == 0)
                t->gran_log2 = 2u;
            else if ((t->chunk_bytes & 1u) == 0)
                t->gran_log2 = 1u;
        }
        /* alpha-style blocks (2 endpoint + 6 index bytes [+ 4 + 4 of the colour half]) go to the field-per-lane
           compressor when everything lines up: default fragment size, whole blocks per chunk, 16-bit streams */
        t->field_period = 0u;
        if (t->compressor == HapCompressorSnappy && frag_log2 == 13u && !ctx->position_lanes && !smaller) {
            if (t->gran_log2 == 1u && (t->format == HapTextureFormat_RGBA_DXT5 || t->format == HapTextureFormat_YCoCg_DXT5) &&
                (t->chunk_bytes & 15u) == 0)
                t->field_period = 4u;
            else if (t->gran_log2 == 2u && (t->format == HapTextureFormat_RGB_DXT1 ||
                                            (t->format == HapTextureFormat_A_RGTC1 && ctx->rgtc1_fields)) && (t->chunk_bytes & 7u) == 0)
                t->field_period = 10u;      /* 2 fields per block, the [4, 4] layout (code 2 | 8) */
            else if (t->gran_log2 == 1u && t->format == HapTextureFormat_A_RGTC1 && ctx->rgtc1_fields == 26u &&
                     (t->chunk_bytes & 7u) == 0 && t->bytes >= ((size_t)2u << 20))
                t->field_period = 2u;       /* [2, 6]: endpoints, indices */
            else if ((flags & HAPGPU_ENCODE_COARSE_MATCHES) && t->gran_log2 == 2u && (t->chunk_bytes & 15u) == 0 &&
                     (t->format == HapTextureFormat_RGBA_BPTC_UNORM || t->format == HapTextureFormat_RGB_BPTC_UNSIGNED_FLOAT ||
                      t->format == HapTextureFormat_RGB_BPTC_SIGNED_FLOAT))
                t->field_period = 12u;      /* opaque 16-byte blocks as four dwords (the size-for-speed option): the block
                                               kernels instead of the position-per-lane ones */
            /* (RGTC1 planes of 2 MiB and more -- block rows longer than a fragment -- take their natural [2, 6] layout:
               same size as the position-per-lane kernel there (0.171 against 0.169 of an 8K alpha plane).  Smaller
               ones stay with positions per lane: their matches start inside the index bytes of the row above, which
               lies inside the fragment -- 0.25 against 0.45 of a 2048 x 512 plane) */
        }
        /* field streams (fragment table version 3): with the table requested, the block compressor's streams come with
           a group table per fragment (96 bytes: where each of the decoder's 64 lanes starts reading) */
        t->half_tiles = ((t->field_period == 4u || t->field_period == 10u || t->field_period == 2u || t->field_period == 12u) && (flags & HAPGPU_ENCODE_FRAGMENT_INDEX) &&
                         !ctx->no_half_tiles) ? 1u : 0u;
        if (t->half_tiles)
            any_half_tiles = 1;
        /* a call that starts from RGBA pictures (hapb_encode_rgba): the block compressor makes the blocks of its
           fragment itself, one pass less over the texture and the pixel loads of one wave under the matching of the
           others (snappy_compress_blocks.hip).  One texture per frame. */
        {
            const HapbBlockEncodeJob *bj = ctx->block_encode_job;
            if (bj && !ctx->no_fusion && count == 1u && (bj->row_bytes & 3u) == 0 &&
                (unsigned long long)bj->row_bytes * bj->height < 0xFFFFFFFFull && bj->width / 4u >= 1u) {
                /* (code: HapGpuTexEnc.reserved bits 24..26; mask bit: which kernel the launcher starts) */
                if (t->field_period == 4u && t->format == HapTextureFormat_YCoCg_DXT5) {
                    fused[i] = 3u; fused_mask |= 1u;
                } else if (t->field_period == 4u && t->format == HapTextureFormat_RGBA_DXT5) {
                    fused[i] = 2u; fused_mask |= 2u;
                }
                /* (DXT1, two blocks to a lane: built and measured -- 115 registers once the match stage's units are read
                   back from the texture instead of kept; 60 4K frames 0.663 against 0.658 ms, 60 1080p frames 0.241
                   against 0.233: its block encoder is further from its instruction-issue limit than the 16-byte ones',
                   nothing to win.  The kernel keeps the two-pass form; the launcher does not start it.) */
            }
        }
        if (t->compressor == HapCompressorSnappy && !fused[i])
            gran_mask |= t->field_period == 4u ? 32u : t->field_period == 10u ? 64u : t->field_period == 2u ? 16u
                         : t->field_period == 12u ? 128u : 1u << t->gran_log2;
        if (t->compressor == HapCompressorSnappy) {
            /* header choice uses the layout that will actually be written (hap.c:425-428) */
            size_t ilen = hapf_instructions_length(t->chunk_count);
            if (flags & HAPGPU_ENCODE_FRAGMENT_INDEX)
                ilen += 8u + (t->half_tiles ? 4u + HAP_GROUP_TABLE_BYTES : 4u) * (size_t)t->chunk_count * t->fpc;
            if (ilen + 4u > 0xFFFFFFu) {          /* instruction container must fit a 24-bit length */
                for (f = 0; f < frame_count; f++) {
                    results[f] = HapResult_Bad_Arguments;
                    output_used[f] = 0;
                }
                return HapResult_Bad_Arguments;
            }
            if (t->bytes + ilen + 4u > 0xFFFFFFu)
                t->header_len = 8u;
        }
        if ((size_t)t->chunk_count * t->fpc > max_frags_per_tex)
            max_frags_per_tex = t->chunk_count * t->fpc;
        frags_per_frame += t->chunk_count * t->fpc;
        chunks_per_frame += t->chunk_count;
        if (t->chunk_count > max_chunks_per_tex)
            max_chunks_per_tex = t->chunk_count;
        stage_in_bytes += align_up(t->bytes, 256);
        frame_raw_bound += t->header_len + t->bytes;
    }
    frame_raw_bound += outer_header;
    /* the block compressor can put the first texture's fragments straight into the frame (no gather pass over them):
       its wavefronts learn the sizes of what lies before them from each other (snappy_compress_blocks.hip) */
    /* (up to 64 chunks: their totals are one load per wavefront.  16 8K frames of 400 chunks: 0.966 ms placed against 0.880
       gathered; of 1 chunk -- 4050 fragments to look back over -- 1.110 against 1.111) */
    /* (frames of two textures -- Hap Q Alpha -- with their FIRST texture placed and the second one gathered behind it:
       measured in round 5 and slower at every chunk count, because such calls run the compressor that reads finished
       textures, whose wavefronts are short: what a placed one waits for -- every fragment before it compressed -- is a
       larger share of its life than in the kernel that makes its blocks itself.  16 8K Hap Q Alpha frames, 24 + 24
       chunks: compress 0.493 -> 0.722 ms for a gather of 0.106 -> 0.036; 4 16K frames, 64 + 64: 0.922 -> 1.537 for
       0.219 -> 0.082.  One texture per frame.) */
    placed = (!ctx->no_placing && !ctx->placing_off && count == 1u && g[0].compressor == HapCompressorSnappy && g[0].field_period != 0u && frag_log2 == 13u &&
              g[0].chunk_count <= 64u) ? 1u : 0u;
    slot_stride = (unsigned)align_up(frag_bytes + frag_bytes / 32u + 64u + HAPGPU_SLOT_SCRATCH_BYTES, 16);
    /* how far placed fragments can reach into the frame buffer if nothing shrinks (the frame is then encoded again, but
       the bytes have been written): the chunked layout's headers and tables + every fragment at its largest (what a slot
       holds).  The bound hap.h asks of the client's buffer covers it except for chunks of a few hundred bytes with the
       private table requested; buffers that do not are served through slots. */
    if (placed) {
        const size_t frags = (size_t)g[0].chunk_count * g[0].fpc;
        size_t ilen = hapf_instructions_length(g[0].chunk_count);
        if (flags & HAPGPU_ENCODE_FRAGMENT_INDEX)
            ilen += 8u + (g[0].half_tiles ? 4u + HAP_GROUP_TABLE_BYTES : 4u) * frags;
        placed_extent = outer_header + 8u + 4u + ilen + 5u * (size_t)g[0].chunk_count + g[0].bytes + frags * (frag_bytes / 32u + 64u);
    }

    /* per-frame checks; frames that fail are left out of the launch */
    live_index = (unsigned *)malloc(sizeof(unsigned) * frame_count);
    stage_off_in = (size_t *)calloc(frame_count, sizeof(size_t));
    stage_off_out = (size_t *)calloc(frame_count, sizeof(size_t));
    if (!live_index || !stage_off_in || !stage_off_out) {
        free(live_index); free(stage_off_in); free(stage_off_out);
        return HapResult_Internal_Error;
    }
    {
        size_t in_total = 0, out_total = 0;
        for (f = 0; f < frame_count; f++) {
            unsigned rc = HapResult_No_Error;
            if (!outputs[f] || output_bytes[f] == 0)
                rc = HapResult_Bad_Arguments;
            for (i = 0; rc == HapResult_No_Error && i < count; i++)
                if (!inputs[(size_t)f * count + i])
                    rc = HapResult_Bad_Arguments;
            if (rc == HapResult_No_Error) {
                /* hap.c:386-389 for the first texture; later textures are re-checked against
                   the space actually left once the sizes are known */
                size_t avail = output_bytes[f] > outer_header ? output_bytes[f] - outer_header : 0;
                if (avail < g[0].bound || output_bytes[f] < frame_raw_bound)
                    rc = HapResult_Buffer_Too_Small;
                /* second texture: the reference compares its bound with what is left after the
                   first section's ACTUAL size (hap.c:589); that size is not known before the
                   launch, so the first section's maximum (raw storage) is assumed */
                else if (count == 2 && avail - (g[0].header_len + g[0].bytes) < g[1].bound)
                    rc = HapResult_Buffer_Too_Small;
            }
            results[f] = rc;
            output_used[f] = 0;
            if (rc != HapResult_No_Error) {
                if (first_error == HapResult_No_Error)
                    first_error = rc;
                continue;
            }
            live_index[live++] = f;
            if (output_bytes[f] < placed_extent)
                placed = 0u;
            if (!inputs_are_device) {
                int staged = 0;
                for (i = 0; i < count; i++)
                    if (!is_dev(ctx, inputs[(size_t)f * count + i]))
                        staged = 1;
                if (staged) {
                    stage_off_in[f] = in_total + 1;         /* +1: 0 means "not staged" */
                    in_total += stage_in_bytes;
                }
            }
            if (!is_dev(ctx, outputs[f])) {
                stage_off_out[f] = out_total + 1;
                /* (+8 per texture: a section written in the chunked form may exceed header + bytes by up to
                   header_len - 1 bytes, see the comparison in frame_pack_kernel / hap.c:478) */
                out_total += align_up(frame_raw_bound + 8u * count > placed_extent ? frame_raw_bound + 8u * count : placed_extent, 256);
            }
        }
        stage_in_bytes = in_total;
        stage_out_bytes = out_total;
    }
    if (live == 0) {
        free(live_index); free(stage_off_in); free(stage_off_out);
        return first_error;
    }
    /* (frames are interleaved over the workgroups: with a dozen of them, few fragments of the same frame are in flight
       together and the waiting costs less than the gather pass; measured on 8K frames: 1 frame 0.130 against 0.114 ms,
       8 the same, 16 0.885 against 0.914, 60 2.90 against 3.08) */
    if (live < ctx->placing_min_frames)
        placed = 0u;
    if (placed && ctx->placing_holdoff) {
        ctx->placing_holdoff -= 1u;
        placed = 0u;
    }
    if (ctx->placing_off)                            /* (after a timeout: counted down call by call, then placing is back) */
        ctx->placing_off -= 1u;

    /* scratch */
    hframes = (HapGpuFrameEnc *)hapgpu_rt_pinned_scratch(rt, P_FRAMES, sizeof(HapGpuFrameEnc) * live);
    dframes = (HapGpuFrameEnc *)hapgpu_rt_device_scratch(rt, D_FRAMES, sizeof(HapGpuFrameEnc) * live);
    dslots = any_snappy ? (uint8_t *)hapgpu_rt_device_scratch(rt, D_SLOTS, (size_t)slot_stride * frags_per_frame * live) : NULL;
    dfragsizes = (uint32_t *)hapgpu_rt_device_scratch(rt, D_FRAGSIZES, sizeof(uint32_t) * (size_t)frags_per_frame * live);
    dcopies = (HapGpuCopyEntry *)hapgpu_rt_device_scratch(rt, D_COPIES, sizeof(HapGpuCopyEntry) * ((size_t)frags_per_frame + chunks_per_frame) * live);
    dpack = hapgpu_rt_device_scratch(rt, D_PACK, (size_t)hapgpu_pack_scratch_bytes_per_chunk() * chunks_per_frame * live);
    if (placed)
        dacc = (uint64_t *)hapgpu_rt_device_scratch(rt, D_CHUNK_ACC, sizeof(uint64_t) * (size_t)chunks_per_frame * live);
    if (any_half_tiles)
        dgrouptables = (uint8_t *)hapgpu_rt_device_scratch(rt, D_GROUPTABLES, (size_t)HAP_GROUP_TABLE_BYTES * frags_per_frame * live);
    if (stage_in_bytes)
        tex_stage = (uint8_t *)hapgpu_rt_device_scratch(rt, D_TEX_STAGE, stage_in_bytes);
    if (stage_out_bytes)
        out_stage = (uint8_t *)hapgpu_rt_device_scratch(rt, D_FRAME_STAGE, stage_out_bytes);
    if (!hframes || !dframes || (placed && !dacc) || (any_snappy && !dslots) || !dfragsizes || !dcopies || !dpack || (any_half_tiles && !dgrouptables) ||
        (stage_in_bytes && !tex_stage) || (stage_out_bytes && !out_stage)) {
        free(live_index); free(stage_off_in); free(stage_off_out);
        for (f = 0; f < frame_count; f++)
            if (results[f] == HapResult_No_Error)
                results[f] = HapResult_Internal_Error;
        return HapResult_Internal_Error;
    }

    /* descriptors (+ host->device staging of textures) */
    {
        unsigned k, rc = 0;
        for (k = 0; k < live; k++) {
            HapGpuFrameEnc *fe = &hframes[k];
            size_t in_cursor = 0;
            f = live_index[k];
            memset(fe, 0, sizeof(*fe));
            fe->dst = (uint64_t)(uintptr_t)(stage_off_out[f] ? out_stage + (stage_off_out[f] - 1) : (uint8_t *)outputs[f]);
            fe->dst_cap = output_bytes[f];
            fe->tex_count = count;
            fe->outer_header_len = outer_header;
            fe->status = HapResult_Internal_Error;   /* overwritten by the pack kernel */
            if (placed)
                fe->chunk_acc = (uint64_t)(uintptr_t)(dacc + (size_t)k * chunks_per_frame);
            if (fused_mask) {
                fe->rgba = ctx->block_encode_job->host_table[f];
                fe->rgba_row_bytes = (uint32_t)ctx->block_encode_job->row_bytes;
                fe->rgba_blocks_x = ctx->block_encode_job->width / 4u;
            }
            for (i = 0; i < count; i++) {
                HapGpuTexEnc *te = &fe->tex[i];
                const void *src = inputs[(size_t)f * count + i];
                if (stage_off_in[f] && !is_dev(ctx, src)) {
                    uint8_t *d = tex_stage + (stage_off_in[f] - 1) + in_cursor;
                    rc |= (unsigned)hapgpu_rt_h2d(rt, d, src, g[i].bytes);
                    src = d;
                }
                in_cursor += align_up(g[i].bytes, 256);
                te->src = (uint64_t)(uintptr_t)src;
                te->bytes = (uint32_t)g[i].bytes;
                te->format_nibble = g[i].nibble;
                te->compressor = g[i].compressor;
                te->chunk_count = g[i].chunk_count;
                te->chunk_bytes = g[i].chunk_bytes;
                te->header_len = g[i].header_len;
                te->frags_per_chunk = g[i].fpc;
                te->frag_first = k * frags_per_frame + (i ? g[0].chunk_count * g[0].fpc : 0u);
                te->emit_index = (flags & HAPGPU_ENCODE_FRAGMENT_INDEX) ? 1u : 0u;
                /* 8 KiB fragments keep hash matches within 3 KiB so that the decoder can halve its ring -- except for
                   small textures, whose block rows are short enough for the row above to lie inside the fragment
                   (rows of up to ~5 KiB: below 1080p for 16-byte blocks, below 4K for 8-byte blocks) */
                {
                    const int small_blocks = g[i].format == HapTextureFormat_RGB_DXT1 || g[i].format == HapTextureFormat_A_RGTC1;
                    /* (field streams with their group table are decoded over a whole-fragment ring: no window) */
                    const int windowed = frag_log2 == 13u && !g[i].half_tiles &&
                                         g[i].bytes >= (small_blocks ? ((size_t)2u << 20) : ((size_t)1u << 20));
                    te->reserved = g[i].gran_log2 | (windowed ? (HAP_FRAGMENT_WINDOW_256 << 8) : 0u) | (g[i].field_period << 16) |
                                   (g[i].half_tiles << 20) | (fused[i] << 24) | ((i == 0u ? placed : 0u) << 27);
                }
            }
        }
        {
            /* The launches of this call as one sequence.  With every buffer in device memory (nothing staged) nothing
               in it depends on the buffers themselves -- their addresses are in the tables copied from pinned memory --
               so the sequence is recorded once per geometry as a HIP graph and replayed with one launch. */
            const HapbBlockEncodeJob *job = ctx->block_encode_job;
            uint64_t key = 0xCBF29CE484222325ull;
            int graph = 2;
            unsigned launch_rc = 0;
#define HAPB_MIX(v) (key = (key ^ (uint64_t)(v)) * 0x100000001B3ull)
            if (stage_in_bytes == 0 && stage_out_bytes == 0 && inputs_are_device != 2 && !smaller && frag_log2 == 13u) {
                HAPB_MIX(live); HAPB_MIX(count); HAPB_MIX(flags); HAPB_MIX(frag_log2); HAPB_MIX(ctx->byte_granular);
                HAPB_MIX(ctx->position_lanes); HAPB_MIX(ctx->rgtc1_fields); HAPB_MIX(ctx->no_half_tiles); HAPB_MIX(fused_mask); HAPB_MIX(placed);
                for (i = 0; i < count; i++) {
                    HAPB_MIX(g[i].format); HAPB_MIX(g[i].compressor); HAPB_MIX(g[i].chunk_count); HAPB_MIX(g[i].bytes);
                }
                if (job) {
                    HAPB_MIX(job->frame_count); HAPB_MIX(job->width); HAPB_MIX(job->height); HAPB_MIX(job->row_bytes);
                    HAPB_MIX(job->wide + 2);
                }
                graph = hapgpu_rt_graph_begin(rt, key);
            }
#undef HAPB_MIX
            /* (a recording that cannot be ended, instantiated or launched has run nothing: graphs are switched off for
               this context and the same launches are issued once more, plainly) */
            while (graph != 1) {
                if (job) {
                    launch_rc |= (unsigned)hapgpu_rt_h2d(rt, job->device_table, job->host_table,
                                                         sizeof(uint64_t) * (size_t)(1u + job->count) * job->frame_count);
                    /* Hap Q Alpha: both textures from one pass over the RGBA (64 + 16 + 8 bytes per block) */
                    if (job->count == 2 && job->formats[0] == HapTextureFormat_YCoCg_DXT5 && job->formats[1] == HapTextureFormat_A_RGTC1)
                        launch_rc |= (unsigned)hapgpu_k_block_encode_batch_ycocg_alpha(rt, job->device_table, job->device_table + job->frame_count,
                                                                                     job->device_table + 2u * (size_t)job->frame_count,
                                                                                     job->frame_count, job->width, job->height,
                                                                                     job->row_bytes, job->wide);
                    else
                        for (i = 0; i < job->count; i++)
                            if (!fused[i])
                                launch_rc |= (unsigned)hapgpu_k_block_encode_batch(rt, job->device_table,
                                                                             job->device_table + (size_t)(1u + i) * job->frame_count,
                                                                             job->frame_count, job->width, job->height, job->row_bytes,
                                                                             job->formats[i], job->wide);
                }
                launch_rc |= (unsigned)hapgpu_rt_h2d(rt, dframes, hframes, sizeof(HapGpuFrameEnc) * live);
                if (placed) {
                    launch_rc |= (unsigned)hapgpu_rt_zero(rt, dacc, sizeof(uint64_t) * (size_t)chunks_per_frame * live);
                    launch_rc |= (unsigned)hapgpu_rt_zero(rt, dfragsizes, sizeof(uint32_t) * (size_t)frags_per_frame * live);
                }
                if (any_snappy)
                    launch_rc |= (unsigned)hapgpu_k_snappy_compress(rt, dframes, live, max_frags_per_tex, frag_log2, dslots, slot_stride,
                                                                    dfragsizes, dgrouptables, gran_mask | (count << 8) | (fused_mask << 16));   /* bits 8..15: textures per frame */
                launch_rc |= (unsigned)hapgpu_k_frame_pack(rt, dframes, live, frag_log2, dslots, slot_stride, dfragsizes, dgrouptables,
                                                           dcopies, frags_per_frame * live, chunks_per_frame, max_chunks_per_tex, count, dpack);
                /* (placed fragments -- single-texture frames -- and their group tables are where they belong: nothing to
                   gather.  A frame reported as not placed is encoded again as a whole.) */
                if (!(placed && count == 1u))
                    launch_rc |= (unsigned)hapgpu_k_frame_gather(rt, dcopies, (frags_per_frame + chunks_per_frame) * live);
                launch_rc |= (unsigned)hapgpu_rt_d2h(rt, hframes, dframes, sizeof(HapGpuFrameEnc) * live);
                if (graph == 0) {
                    /* (a recording can also be invalidated from outside -- another thread of the process synchronising the
                       device while it runs: then the launches themselves report errors.  Either way nothing has run.) */
                    const int ended = hapgpu_rt_graph_end(rt, key, launch_rc != 0);
                    if (ended != 0 || launch_rc != 0) {
                        hapgpu_rt_graphs_disable(rt);
                        graph = 2;
                        launch_rc = 0;
                        continue;
                    }
                }
                break;
            }
            rc |= launch_rc;
        }
        /* everything the call launches is on the stream: the rest -- waiting, reading the results, the second pass over
           frames that were not placed -- is hapb_encode_complete's, at once or (HapGpuEncodeFramesRGBABegin / ...Finish)
           when the client asks for the results */
        {
            HapbEncodePending *pd = (HapbEncodePending *)calloc(1, sizeof(*pd));
            const size_t n_in = (size_t)frame_count * count;
            if (pd) {
                pd->inputs = (const void **)malloc(sizeof(void *) * n_in);
                pd->outputs = (void **)malloc(sizeof(void *) * frame_count);
                pd->output_bytes = (unsigned long *)malloc(sizeof(unsigned long) * frame_count);
            }
            if (!pd || !pd->inputs || !pd->outputs || !pd->output_bytes) {
                if (pd) { free(pd->inputs); free(pd->outputs); free(pd->output_bytes); }
                free(pd);
                hapgpu_rt_sync(rt);
                for (k = 0; k < live; k++)
                    results[live_index[k]] = HapResult_Internal_Error;
                free(live_index); free(stage_off_in); free(stage_off_out);
                return HapResult_Internal_Error;
            }
            memcpy(pd->inputs, inputs, sizeof(void *) * n_in);
            memcpy(pd->outputs, outputs, sizeof(void *) * frame_count);
            memcpy(pd->output_bytes, output_bytes, sizeof(unsigned long) * frame_count);
            for (i = 0; i < count; i++) {
                pd->input_bytes[i] = input_bytes[i];
                pd->formats[i] = formats[i];
                pd->compressors[i] = compressors[i];
                pd->chunk_counts[i] = chunk_counts[i];
            }
            pd->frame_count = frame_count;
            pd->count = count;
            pd->flags = flags;
            pd->inputs_are_device = inputs_are_device;
            pd->output_used = output_used;
            pd->results = results;
            pd->live = live;
            pd->placed = placed;
            pd->first_error = first_error;
            pd->launch_rc = rc;
            pd->live_index = live_index;
            pd->stage_off_out = stage_off_out;
            pd->hframes = hframes;
            pd->out_stage = out_stage;
            if (ctx->block_encode_job) {
                pd->job = *ctx->block_encode_job;
                pd->has_job = 1;
            }
            free(stage_off_in);
            if (ctx->defer_encode) {
                ctx->pending_encode = pd;
                return first_error;
            }
            return hapb_encode_complete(ctx, pd);
        }
    }
}

/* The second half of hapb_encode: waits for the launches, reads the per-frame results back, copies staged frames to the
   client's host buffers, and encodes the frames that were not placed once more.  Frees `pd`. */
unsigned hapb_encode_complete(HapGpuContext *ctx, HapbEncodePending *pd)
{
    hapgpu_rt *rt = ctx->rt;
    const unsigned frame_count = pd->frame_count, count = pd->count, live = pd->live, flags = pd->flags;
    unsigned *const results = pd->results;
    unsigned long *const output_used = pd->output_used;
    unsigned first_error = pd->first_error, rc = pd->launch_rc, f, i, k;
    rc |= (unsigned)hapgpu_rt_sync(rt);
    if (rc) {
        for (k = 0; k < live; k++)
            results[pd->live_index[k]] = HapResult_Internal_Error;
        first_error = HapResult_Internal_Error;
        goto done;
    }
    /* results (+ device->host copy of staged frames) */
    {
        int copied = 0;
        for (k = 0; k < live; k++) {
            HapGpuFrameEnc *fe = &pd->hframes[k];
            f = pd->live_index[k];
            results[f] = fe->status;
            if (fe->status == HAPGPU_STATUS_NOT_PLACED && (fe->reserved & 1u)) {
                /* a wavefront gave up waiting for its predecessors (snappy_compress_blocks.hip): something keeps the
                   grid from advancing in order -- this context gathers from now on */
                ctx->placement_timeouts += 1u;
                if (!ctx->placing_off)
                    fprintf(stderr, "hap_amd: a placed encode waited too long for its predecessors (another kernel, a profiler, CU "
                                    "masking?): this context gathers for its next 64 encode calls\n");
                ctx->placing_off = 64u;              /* calls that gather before placing is tried again */
            }
            if (fe->status == HapResult_No_Error) {
                output_used[f] = (unsigned long)fe->bytes_used;
                if (pd->stage_off_out[f]) {
                    if (hapgpu_rt_d2h(rt, pd->outputs[f], pd->out_stage + (pd->stage_off_out[f] - 1), (size_t)fe->bytes_used))
                        results[f] = HapResult_Internal_Error;
                    copied = 1;
                }
            }
            if (results[f] != HapResult_No_Error && results[f] != HAPGPU_STATUS_NOT_PLACED && first_error == HapResult_No_Error)
                first_error = results[f];
        }
        if (copied && hapgpu_rt_sync(rt))
            first_error = HapResult_Internal_Error;
    }
    /* frames with a chunk that Snappy did not shrink (stored raw, hap.c:460-466: everything behind it lies elsewhere
       than the wavefronts assumed): once more, through slots.  The textures are where they were -- the client's --
       or, in a call that started from pictures, are made again from the pictures: the fused kernel of a placed call
       does not keep them (snappy_compress_blocks.hip). */
    if (pd->placed) {
        unsigned again = 0;
        for (f = 0; f < frame_count; f++)
            again += results[f] == HAPGPU_STATUS_NOT_PLACED;
        if (again) {
            /* all of them in one batch */
            const HapbBlockEncodeJob *const outer_job = ctx->block_encode_job;
            const unsigned saved_no_placing = ctx->no_placing, saved_defer = ctx->defer_encode;
            HapbBlockEncodeJob rjob;
            uint64_t *rtable = NULL;
            const void **rin = (const void **)malloc(sizeof(void *) * (size_t)again * count);
            void **rout = (void **)malloc(sizeof(void *) * again);
            unsigned long *rcap = (unsigned long *)malloc(sizeof(unsigned long) * again * 2u);
            unsigned *rres = (unsigned *)malloc(sizeof(unsigned) * again * 2u);
            if (pd->has_job)
                rtable = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(1u + pd->job.count) * again);
            if (rin && rout && rcap && rres && (!pd->has_job || rtable)) {
                unsigned long *rused = rcap + again;
                unsigned *rmap = rres + again, m = 0;
                for (f = 0; f < frame_count; f++)
                    if (results[f] == HAPGPU_STATUS_NOT_PLACED) {
                        for (i = 0; i < count; i++)
                            rin[(size_t)m * count + i] = pd->inputs[(size_t)f * count + i];
                        rout[m] = pd->outputs[f];
                        rcap[m] = pd->output_bytes[f];
                        rres[m] = HapResult_Internal_Error;
                        rused[m] = 0;
                        if (pd->has_job) {
                            /* the pictures and texture places of these frames, as a table of its own */
                            unsigned t;
                            for (t = 0; t < 1u + pd->job.count; t++)
                                rtable[(size_t)t * again + m] = pd->job.host_table[(size_t)t * pd->job.frame_count + f];
                        }
                        rmap[m++] = f;
                    }
                ctx->block_encode_job = NULL;
                if (pd->has_job) {
                    /* the table goes where the first pass's was: pinned memory at an address that stays -- a recorded
                       launch sequence (HAP_AMD_GRAPHS) copies from THAT address again when it is replayed (found by
                       tools/stress.py: a table in malloc'd memory was gone by then) */
                    uint64_t *pinned = (uint64_t *)hapgpu_rt_pinned_scratch(rt, P_BC_PTRS, sizeof(uint64_t) * (size_t)(1u + pd->job.count) * again);
                    rjob = pd->job;
                    rjob.frame_count = again;
                    if (pinned) {
                        memcpy(pinned, rtable, sizeof(uint64_t) * (size_t)(1u + pd->job.count) * again);
                        rjob.host_table = pinned;
                    } else {
                        rjob.host_table = rtable;
                    }
                    ctx->block_encode_job = &rjob;
                }
                ctx->no_placing = 1u;
                ctx->defer_encode = 0u;
                hapb_encode(ctx, again, count, rin, pd->input_bytes, pd->formats, pd->compressors, pd->chunk_counts, rout, rcap, rused, rres,
                            flags, pd->inputs_are_device);
                ctx->defer_encode = saved_defer;
                ctx->no_placing = saved_no_placing;
                ctx->block_encode_job = outer_job;
                for (m = 0; m < again; m++) {
                    /* (a call without placing cannot report a frame as not placed) */
                    results[rmap[m]] = rres[m] == HAPGPU_STATUS_NOT_PLACED ? HapResult_Internal_Error : rres[m];
                    output_used[rmap[m]] = rused[m];
                }
            } else {
                for (f = 0; f < frame_count; f++)
                    if (results[f] == HAPGPU_STATUS_NOT_PLACED)
                        results[f] = HapResult_Internal_Error;
            }
            free(rin); free(rout); free(rcap); free(rres); free(rtable);
            ctx->placement_retries += again;
            for (f = 0; f < frame_count; f++)
                if (results[f] != HapResult_No_Error && first_error == HapResult_No_Error)
                    first_error = results[f];
            /* content that does not shrink tends to stay: a call that had to encode most of its frames twice
               keeps the next calls from trying (and then tries again) */
            if (2u * again > live)
                ctx->placing_holdoff = ctx->placing_holdoff_calls;
        }
    }
done:
    free(pd->live_index); free(pd->stage_off_out);
    free(pd->inputs); free(pd->outputs); free(pd->output_bytes);
    free(pd);
    return first_error;
}

void hapb_encode_abandon(HapGpuContext *ctx, HapbEncodePending *pd)
{
    (void)hapgpu_rt_sync(ctx->rt);
    free(pd->live_index); free(pd->stage_off_out);
    free(pd->inputs); free(pd->outputs); free(pd->output_bytes);
    free(pd);
}

unsigned hapb_compress_rgba(HapGpuContext *ctx, const void *rgba, unsigned width, unsigned height,
                            unsigned long row_bytes, unsigned format, void *output,
                            unsigned long output_bytes, unsigned long *used, int synchronise)
{
    hapgpu_rt *rt = ctx->rt;
    size_t block = (format == HapTextureFormat_RGB_DXT1 || format == HapTextureFormat_A_RGTC1) ? 8u : 16u;
    size_t need, rgba_bytes;
    const void *src = rgba;
    void *dst = output;
    int rc = 0;
    if (context_busy(ctx, NULL, 0))
        return HapResult_Internal_Error;
    if (!rgba || !output || width == 0 || height == 0 || (width & 3u) || (height & 3u) ||
        row_bytes < (unsigned long)width * 4ul ||
        (format != HapTextureFormat_RGB_DXT1 && format != HapTextureFormat_RGBA_DXT5 &&
         format != HapTextureFormat_YCoCg_DXT5 && format != HapTextureFormat_A_RGTC1))
        return HapResult_Bad_Arguments;
    need = (size_t)(width / 4u) * (height / 4u) * block;
    if (output_bytes < need)
        return HapResult_Buffer_Too_Small;
    rgba_bytes = (size_t)row_bytes * (height - 1u) + (size_t)width * 4u;
    if (!is_dev(ctx, rgba)) {
        void *s = hapgpu_rt_device_scratch(rt, D_RGBA_STAGE, rgba_bytes);
        if (!s || hapgpu_rt_h2d(rt, s, rgba, rgba_bytes))
            return HapResult_Internal_Error;
        src = s;
    }
    if (!is_dev(ctx, output)) {
        dst = hapgpu_rt_device_scratch(rt, D_BC_TEX, need);
        if (!dst)
            return HapResult_Internal_Error;
    }
    rc = hapgpu_k_block_encode(rt, src, width, height, row_bytes, format, dst);
    if (rc == 1)
        return HapResult_Bad_Arguments;
    if (rc)
        return HapResult_Internal_Error;
    if (dst != output && hapgpu_rt_d2h(rt, output, dst, need))
        return HapResult_Internal_Error;
    if ((synchronise || dst != output) && hapgpu_rt_sync(rt))
        return HapResult_Internal_Error;
    if (used)
        *used = (unsigned long)need;
    return HapResult_No_Error;
}

unsigned hapb_decompress_rgba(HapGpuContext *ctx, const void *texture, unsigned long texture_bytes, unsigned format,
                              const void *alpha, unsigned long alpha_bytes, unsigned width, unsigned height,
                              void *rgba, unsigned long row_bytes)
{
    hapgpu_rt *rt = ctx->rt;
    size_t block = format == HapTextureFormat_RGB_DXT1 ? 8u : 16u;
    size_t need, alpha_need, rgba_bytes;
    const void *src = texture, *asrc = alpha;
    void *dst = rgba;
    int rc;
    if (context_busy(ctx, NULL, 0))
        return HapResult_Internal_Error;
    if (!texture || !rgba || width == 0 || height == 0 || (width & 3u) || (height & 3u) ||
        row_bytes < (unsigned long)width * 4ul ||
        (format != HapTextureFormat_RGB_DXT1 && format != HapTextureFormat_RGBA_DXT5 &&
         format != HapTextureFormat_YCoCg_DXT5))
        return HapResult_Bad_Arguments;
    need = (size_t)(width / 4u) * (height / 4u) * block;
    alpha_need = (size_t)(width / 4u) * (height / 4u) * 8u;
    if (texture_bytes < need || (alpha && alpha_bytes < alpha_need))
        return HapResult_Bad_Arguments;
    rgba_bytes = (size_t)row_bytes * (height - 1u) + (size_t)width * 4u;
    if (!is_dev(ctx, texture)) {
        void *s = hapgpu_rt_device_scratch(rt, D_BC_TEX, need + alpha_need + 256);
        if (!s || hapgpu_rt_h2d(rt, s, texture, need))
            return HapResult_Internal_Error;
        src = s;
        if (alpha && !is_dev(ctx, alpha)) {
            uint8_t *a = (uint8_t *)s + align_up(need, 256);
            if (hapgpu_rt_h2d(rt, a, alpha, alpha_need))
                return HapResult_Internal_Error;
            asrc = a;
        }
    } else if (alpha && !is_dev(ctx, alpha)) {
        void *a = hapgpu_rt_device_scratch(rt, D_BC_TEX, alpha_need);
        if (!a || hapgpu_rt_h2d(rt, a, alpha, alpha_need))
            return HapResult_Internal_Error;
        asrc = a;
    }
    if (!is_dev(ctx, rgba)) {
        dst = hapgpu_rt_device_scratch(rt, D_RGBA_STAGE, rgba_bytes);
        if (!dst)
            return HapResult_Internal_Error;
    }
    rc = hapgpu_k_block_decode(rt, src, asrc, width, height, format, dst, row_bytes);
    if (rc == 1)
        return HapResult_Bad_Arguments;
    if (rc)
        return HapResult_Internal_Error;
    if (dst != rgba && hapgpu_rt_d2h(rt, rgba, dst, rgba_bytes))
        return HapResult_Internal_Error;
    if (hapgpu_rt_sync(rt))
        return HapResult_Internal_Error;
    return HapResult_No_Error;
}

unsigned hapb_encode_rgba(HapGpuContext *ctx, unsigned frame_count, const void *const *rgba_frames,
                          unsigned width, unsigned height, unsigned long row_bytes, unsigned count,
                          const unsigned *formats, const unsigned *compressors, const unsigned *chunk_counts,
                          void *const *outputs, const unsigned long *output_bytes,
                          unsigned long *output_used, unsigned *results, unsigned flags)
{
    hapgpu_rt *rt = ctx->rt;
    unsigned long tex_bytes[2] = {0, 0};
    size_t per_frame = 0, rgba_bytes, tex_off[2] = {0, 0};
    unsigned i, f, rc;
    uint8_t *textures, *rgba_stage = NULL;
    const void **tex_ptrs;
    uint64_t *hsrc, *dsrc;
    int wide = 1;
    if (frame_count == 0)
        return HapResult_No_Error;
    if (context_busy(ctx, results, frame_count))
        return HapResult_Internal_Error;
    if (!results || !rgba_frames || count == 0 || count > 2 || !formats || width == 0 || height == 0 ||
        (width & 3u) || (height & 3u) || row_bytes < (unsigned long)width * 4ul) {
        for (f = 0; results && f < frame_count; f++)
            results[f] = HapResult_Bad_Arguments;
        return HapResult_Bad_Arguments;
    }
    for (i = 0; i < count; i++) {
        size_t block;
        if (formats[i] != HapTextureFormat_RGB_DXT1 && formats[i] != HapTextureFormat_RGBA_DXT5 &&
            formats[i] != HapTextureFormat_YCoCg_DXT5 && formats[i] != HapTextureFormat_A_RGTC1) {
            for (f = 0; f < frame_count; f++)
                results[f] = HapResult_Bad_Arguments;
            return HapResult_Bad_Arguments;
        }
        block = (formats[i] == HapTextureFormat_RGB_DXT1 || formats[i] == HapTextureFormat_A_RGTC1) ? 8u : 16u;
        tex_bytes[i] = (unsigned long)((size_t)(width / 4u) * (height / 4u) * block);
        tex_off[i] = per_frame;
        per_frame += align_up(tex_bytes[i], 256);
    }
    rgba_bytes = (size_t)row_bytes * (height - 1u) + (size_t)width * 4u;
    textures = (uint8_t *)hapgpu_rt_device_scratch(rt, D_BC_TEX, per_frame * frame_count);
    tex_ptrs = (const void **)malloc(sizeof(void *) * (size_t)frame_count * count);
    /* address table of the batched block-encode launches: [sources][outputs of texture 0][of texture 1] */
    hsrc = (uint64_t *)hapgpu_rt_pinned_scratch(rt, P_BC_PTRS, sizeof(uint64_t) * 3u * frame_count);
    dsrc = (uint64_t *)hapgpu_rt_device_scratch(rt, D_BC_PTRS, sizeof(uint64_t) * 3u * frame_count);
    if (!textures || !tex_ptrs || !hsrc || !dsrc) {
        free(tex_ptrs);
        for (f = 0; f < frame_count; f++)
            results[f] = HapResult_Internal_Error;
        return HapResult_Internal_Error;
    }
    memset(hsrc, 0, sizeof(uint64_t) * 3u * frame_count);       /* address 0: the kernel skips the picture */
    for (f = 0; f < frame_count; f++) {
        const void *src = rgba_frames[f];
        for (i = 0; i < count; i++)
            tex_ptrs[(size_t)f * count + i] = NULL;     /* NULL input => Bad_Arguments for that frame */
        if (!src)
            continue;
        if (!is_dev(ctx, src)) {
            /* one staging buffer per frame so that uploads and kernels can overlap on the stream */
            if (!rgba_stage) {
                rgba_stage = (uint8_t *)hapgpu_rt_device_scratch(rt, D_RGBA_STAGE, align_up(rgba_bytes, 256) * frame_count);
                if (!rgba_stage)
                    break;
            }
            if (hapgpu_rt_h2d(rt, rgba_stage + align_up(rgba_bytes, 256) * f, src, rgba_bytes))
                break;
            src = rgba_stage + align_up(rgba_bytes, 256) * f;
        }
        hsrc[f] = (uint64_t)(uintptr_t)src;
        if (((uintptr_t)src | row_bytes) & 15u)
            wide = 0;
        for (i = 0; i < count; i++) {
            uint8_t *t = textures + per_frame * f + tex_off[i];
            hsrc[(size_t)(1u + i) * frame_count + f] = (uint64_t)(uintptr_t)t;
            tex_ptrs[(size_t)f * count + i] = t;
        }
    }
    /* the block encode (one launch per texture format over the whole batch, addresses in a small device table) is
       issued by hapb_encode at the head of its own launches: one sequence per call */
    {
        HapbBlockEncodeJob job;
        memset(&job, 0, sizeof(job));
        job.host_table = hsrc;
        job.device_table = dsrc;
        job.frame_count = frame_count;
        job.count = count;
        job.width = width;
        job.height = height;
        job.row_bytes = row_bytes;
        job.wide = wide;
        for (i = 0; i < count; i++)
            job.formats[i] = formats[i];
        ctx->block_encode_job = &job;
        rc = hapb_encode(ctx, frame_count, count, tex_ptrs, tex_bytes, formats, compressors, chunk_counts, outputs,
                         output_bytes, output_used, results, flags, rgba_stage ? 2 : 1);
        ctx->block_encode_job = NULL;
    }
    free(tex_ptrs);
    return rc;
}

/* ================================================================== decode */

/* the block layout ("fields per block" code of the fragment table) a texture format's field streams have; 0: none */
static unsigned field_layout_of_format(unsigned format)
{
    switch (format) {
    case HapTextureFormat_RGBA_DXT5:
    case HapTextureFormat_YCoCg_DXT5: return 4u;
    case HapTextureFormat_RGB_DXT1: return 2u;
    case HapTextureFormat_A_RGTC1: return 6u;
    default: return 0u;
    }
}

typedef struct fetch_ctx {
    HapGpuContext *ctx;
    const uint8_t *device_frame;
} fetch_ctx;

static int fetch_from_device(void *user, uint64_t offset, uint64_t length, uint8_t *dst)
{
    fetch_ctx *fc = (fetch_ctx *)user;
    if (hapgpu_rt_d2h(fc->ctx->rt, dst, fc->device_frame + offset, (size_t)length))
        return 1;
    return hapgpu_rt_sync(fc->ctx->rt);
}

/* work function handed to the client's HapDecodeCallback: marks chunk `index` as requested */
typedef struct request_marks {
    unsigned count;
    volatile unsigned char *requested;
} request_marks;

static void mark_chunk(void *p, unsigned index)
{
    request_marks *m = (request_marks *)p;
    if (m && index < m->count)
        m->requested[index] = 1;
}

/* Unit slots to reserve behind a whole-stream unit for the block scan (snappy_decode.hip): one per 64 KiB of output.
   The output length is on the device (the stream's varint); bound it by the client's buffer and by the format's
   largest expansion (a 3-byte copy element produces 64 bytes).  Streams of one block need none. */
static void scan_entry(HapGpuScanChunk *e, unsigned unit, unsigned long src_len, unsigned long dst_cap, uint32_t *dbpos,
                       unsigned *seg_cursor, unsigned *word_cursor, unsigned seg_bytes, unsigned *fine_cursor);

static unsigned stream_scan_segments(unsigned long src_len, unsigned seg_bytes)
{
    return (unsigned)(((unsigned long long)src_len + 15u + seg_bytes - 1u) / seg_bytes);
}

static unsigned long long stream_output_bound(unsigned long src_len, unsigned long dst_cap)
{
    unsigned long long bound = (unsigned long long)src_len * 22u;
    return bound > dst_cap ? dst_cap : bound;
}

/* 64 KiB blocks (libsnappy's) */
static unsigned stream_coarse_slots(unsigned long src_len, unsigned long dst_cap)
{
    unsigned long long bound = stream_output_bound(src_len, dst_cap);
    if (bound <= 65536u)
        return 0u;
    bound = (bound + 65535u) / 65536u;
    return bound > 4096u ? 4096u : (unsigned)bound;
}

/* 8 KiB blocks (the fragments of this library's own table-less streams): none for streams that are not scanned, or
   too long to give every 8 KiB a slot */
static unsigned stream_fine_slots(unsigned long src_len, unsigned long dst_cap)
{
    unsigned long long bound = stream_output_bound(src_len, dst_cap);
    if (bound <= 65536u)
        return 0u;
    bound = (bound + HAPGPU_SCAN_FINE - 1u) / HAPGPU_SCAN_FINE;
    return bound > 32768u ? 0u : (unsigned)bound;
}

/* unit slots directly behind a whole-stream unit: one per 64 KiB block (the 8 KiB blocks' slots lie behind all the
   ordinary units of the call) */
static unsigned stream_block_slots(unsigned long src_len, unsigned long dst_cap)
{
    return stream_coarse_slots(src_len, dst_cap);
}

/* words of block positions a scanned stream needs: one per mark + the end */
static unsigned stream_mark_words(unsigned long src_len, unsigned long dst_cap)
{
    const unsigned coarse = stream_coarse_slots(src_len, dst_cap), fine = stream_fine_slots(src_len, dst_cap);
    return (fine > coarse ? fine : coarse) + 1u;
}

static void scan_entry(HapGpuScanChunk *e, unsigned unit, unsigned long src_len, unsigned long dst_cap, uint32_t *dbpos,
                       unsigned *seg_cursor, unsigned *word_cursor, unsigned seg_bytes, unsigned *fine_cursor)
{
    const unsigned coarse = stream_coarse_slots(src_len, dst_cap), fine = coarse ? stream_fine_slots(src_len, dst_cap) : 0u;
    memset(e, 0, sizeof(*e));
    e->unit = unit;
    e->seg_first = *seg_cursor;
    e->seg_count = stream_scan_segments(src_len, seg_bytes);
    e->seg_bytes = seg_bytes;
    e->slots = coarse;
    e->fine_slots = fine;              /* (its words of block positions; the unit slots come from the call's pool, on the device) */
    e->fine_unit_first = 0;
    e->bpos = (uint64_t)(uintptr_t)(dbpos + *word_cursor);
    *seg_cursor += e->seg_count;
    *word_cursor += stream_mark_words(src_len, dst_cap);
    *fine_cursor += fine;
}

unsigned hapb_decode(HapGpuContext *ctx, unsigned frame_count, const void *const *inputs,
                     const unsigned long *input_bytes, unsigned index, void *const *outputs,
                     const unsigned long *output_bytes, unsigned long *output_used,
                     unsigned *output_formats, unsigned *results, unsigned flags,
                     HapDecodeCallback callback, void *callback_info)
{
    hapgpu_rt *rt = ctx->rt;
    hapf_texture_plan *plans;
    hapf_reader *readers;
    fetch_ctx *fetchers;
    unsigned f, live = 0, first_error = HapResult_No_Error, total_units = 0, total_chunks = 0;
    unsigned frag_log2_seen = 0, frag_kinds = 0, max_chunks = 0, max_stream_src = 0, guess_units = 0;
    unsigned char *guess_frame = NULL;                         /* frames whose chunks may be field-stream fragments without a table */
    int use_guess = 0, use_scan_guess = 0;
    uint8_t *dguess = NULL;
    int any_stream = 0, need_retry = 0;
    unsigned scan_chunks = 0, scan_segs = 0, scan_words = 0;   /* streams with BLOCK slots; their segments; bpos words */
    /* (r06 measured shorter segments for calls of few streams -- a wavefront of the scan's first kernel walks a segment's
       windows one after the other -- and found nothing: one reference-made 8K frame, walk + merge 56 + 80 us at 2 KiB,
       60 + 122 at 1 KiB, the whole scan 0.088 ms at 2 KiB and at 4 KiB) */
    const unsigned scan_seg_bytes = HAPGPU_SCAN_SEGMENT;
    unsigned fine_total = 0;                                   /* unit slots for the 8 KiB blocks of scanned streams */
    uint32_t *dwork = NULL;                                    /* [0]: count, then the fine units the scan listed */
    unsigned far_seen = 0;
    /* (a call for several textures of the same frames: HapGpuDecodeFrameTextures hands the per-entry indices over) */
    const unsigned *const entry_index = ctx->decode_indices;
#define TEXTURE_INDEX(f) (entry_index ? entry_index[f] : index)
    const int block_scan = !(flags & HAPGPU_DECODE_NO_BLOCK_SCAN) && !ctx->no_block_scan;
    uint8_t *prefix = NULL, *in_stage = NULL, *out_stage = NULL;
    size_t in_stage_bytes = 0, out_stage_bytes = 0, prefix_bytes = PREFIX_BYTES;
    unsigned prefix_pass;
    size_t *in_off, *out_off;
    HapGpuDecodeJob *hjobs, *djobs;
    HapGpuChunkIn *hchunks, *dchunks;
    HapGpuDecodeUnit *dunits;
    HapGpuScanChunk *hscan = NULL, *dscan = NULL;
    HapGpuScanSegment *dsegs = NULL;
    uint32_t *dbpos = NULL;
    uint8_t *drecs = NULL, *djoins = NULL;
    unsigned *job_of_frame;
    unsigned char *in_dev = NULL, *out_dev = NULL;     /* pointer classification, done once per buffer */
    unsigned char *client_marks = NULL;                /* what the client's callback asked for (single-frame path) */
    unsigned client_marks_count = 0;
    int rc = 0;

#ifdef HAPB_TRACE
    double hapb_t0 = hapb_now_us();
#endif
    ctx->decode_indices = NULL;
    if (frame_count == 0)
        return HapResult_No_Error;
    if (!results)
        return HapResult_Bad_Arguments;
    if (context_busy(ctx, results, frame_count))
        return HapResult_Internal_Error;
    if (!inputs || !input_bytes || !outputs || !output_bytes || index > 1) {
        for (f = 0; f < frame_count; f++)
            results[f] = HapResult_Bad_Arguments;
        return HapResult_Bad_Arguments;
    }
    plans = (hapf_texture_plan *)calloc(frame_count, sizeof(*plans));
    readers = (hapf_reader *)calloc(frame_count, sizeof(*readers));
    fetchers = (fetch_ctx *)calloc(frame_count, sizeof(*fetchers));
    in_off = (size_t *)calloc(frame_count, sizeof(size_t));
    out_off = (size_t *)calloc(frame_count, sizeof(size_t));
    job_of_frame = (unsigned *)calloc(frame_count, sizeof(unsigned));
    in_dev = (unsigned char *)calloc(frame_count, 1);
    out_dev = (unsigned char *)calloc(frame_count, 1);
    guess_frame = (unsigned char *)calloc(frame_count, 1);
    if (!plans || !readers || !fetchers || !in_off || !out_off || !job_of_frame || !in_dev || !out_dev || !guess_frame) {
        rc = 1;
        goto fail_alloc;
    }

    /* 1. make the container headers host-visible: device frames get a prefix copied back */
    {
        unsigned device_frames = 0;
        for (f = 0; f < frame_count; f++) {
            results[f] = HapResult_No_Error;
            if (!inputs[f] || !outputs[f]) {
                results[f] = HapResult_Bad_Arguments;
                continue;
            }
            in_dev[f] = (unsigned char)is_dev(ctx, inputs[f]);
            out_dev[f] = (unsigned char)is_dev(ctx, outputs[f]);
            if (in_dev[f])
                device_frames++;
        }
        /* (a second pass with a longer prefix when the tables of some frame do not fit the first one -- frames of
           thousands of chunks, HAPGPU_ENCODE_FINE_CHUNKS: one more round trip for the whole batch instead of a fetch
           on demand per frame and table) */
        for (prefix_pass = 0; prefix_pass < 2u; prefix_pass++) {
            size_t need_max = 0;
            if (device_frames && prefix_pass == 1u) {
                /* the longer prefix is a convenience, never a reason to fail the batch: if its scratch cannot
                   be had, the call goes on with the short prefixes it already holds the sizes for, and tables beyond them
                   are fetched on demand, frame by frame */
                const size_t block = (size_t)prefix_bytes * frame_count;
                if (!hapgpu_rt_device_scratch(rt, D_PREFIX, 2u * block + sizeof(uint64_t) * frame_count) ||
                    !hapgpu_rt_pinned_scratch(rt, P_PREFIX, 2u * block + sizeof(uint64_t) * frame_count))
                    prefix_bytes = PREFIX_BYTES;
            }
            if (device_frames) {
                /* one gather kernel + one copy bring every device frame's header prefix to the host -- and, for a later
                   texture of a multi-texture frame (which begins where the first one ends, far beyond the prefix), the
                   bytes at its section too: the kernel reads the two section headers in front of it itself.
                   Device block: prefixes | second prefixes | their offsets;  upload: pointers | lengths | wanted flags */
                const size_t up_bytes = sizeof(uint64_t) * 2u * frame_count + frame_count;
                const size_t block = (size_t)prefix_bytes * frame_count;
                uint64_t *hptr = (uint64_t *)hapgpu_rt_pinned_scratch(rt, P_PTRS, up_bytes);
                uint64_t *dptr = (uint64_t *)hapgpu_rt_device_scratch(rt, D_PTRS, up_bytes);
                uint8_t *dprefix = (uint8_t *)hapgpu_rt_device_scratch(rt, D_PREFIX, 2u * block + sizeof(uint64_t) * frame_count);
                unsigned far = 0;
                prefix = (uint8_t *)hapgpu_rt_pinned_scratch(rt, P_PREFIX, 2u * block + sizeof(uint64_t) * frame_count);
                if (!prefix || !hptr || !dptr || !dprefix) {
                    rc = 1;
                    goto fail_alloc;
                }
                for (f = 0; f < frame_count; f++) {
                    const int dev_frame = results[f] == HapResult_No_Error && in_dev[f];
                    hptr[f] = dev_frame ? (uint64_t)(uintptr_t)inputs[f] : 0u;
                    hptr[frame_count + f] = input_bytes[f];
                    ((uint8_t *)(hptr + 2u * frame_count))[f] = (uint8_t)(dev_frame && TEXTURE_INDEX(f) > 0);
                    far += dev_frame && TEXTURE_INDEX(f) > 0;
                }
                if (hapgpu_rt_pinned_is_mapped(rt)) {
                    /* the kernel reads the pointers from, and writes the prefixes to, the pinned buffers themselves: one
                       launch instead of a copy up, a launch and a copy back */
                    if (far)
                        rc |= hapgpu_k_gather_prefixes_far(rt, hptr, hptr + frame_count, frame_count, prefix_bytes, prefix,
                                                           (const uint8_t *)(hptr + 2u * frame_count), prefix + block,
                                                           (uint64_t *)(prefix + 2u * block));
                    else
                        rc |= hapgpu_k_gather_prefixes(rt, hptr, hptr + frame_count, frame_count, prefix_bytes, prefix);
                } else {
                    rc |= hapgpu_rt_h2d(rt, dptr, hptr, up_bytes);
                    if (far) {
                        rc |= hapgpu_k_gather_prefixes_far(rt, dptr, dptr + frame_count, frame_count, prefix_bytes, dprefix,
                                                           (const uint8_t *)(dptr + 2u * frame_count), dprefix + block,
                                                           (uint64_t *)(dprefix + 2u * block));
                        rc |= hapgpu_rt_d2h(rt, prefix, dprefix, 2u * block + sizeof(uint64_t) * frame_count);
                    } else {
                        rc |= hapgpu_k_gather_prefixes(rt, dptr, dptr + frame_count, frame_count, prefix_bytes, dprefix);
                        rc |= hapgpu_rt_d2h(rt, prefix, dprefix, block);
                    }
                }
                far_seen = far;
            }
            for (f = 0; f < frame_count; f++) {
                if (results[f] != HapResult_No_Error)
                    continue;
                if (in_dev[f]) {
                    size_t n = input_bytes[f] < prefix_bytes ? input_bytes[f] : prefix_bytes;
                    hapf_reader_init_host(&readers[f], prefix + (size_t)prefix_bytes * f, n);
                    fetchers[f].ctx = ctx;
                    fetchers[f].device_frame = (const uint8_t *)inputs[f];
                    readers[f].fetch = fetch_from_device;
                    readers[f].user = &fetchers[f];
                    readers[f].total_len = input_bytes[f];
                } else {
                    hapf_reader_init_host(&readers[f], inputs[f], input_bytes[f]);
                    in_off[f] = in_stage_bytes + 1;
                    in_stage_bytes += align_up(input_bytes[f], 256);
                }
            }
            if (device_frames)
                rc |= hapgpu_rt_sync(rt);
            if (rc)
                goto fail_alloc;
            if (far_seen) {
                /* the second prefix is a second window of the reader -- taken only where the host, reading the same two
                   section headers, arrives at the offset the kernel reported (anything else: the reader fetches on demand) */
                const size_t block = (size_t)prefix_bytes * frame_count;
                const uint64_t *far_at = (const uint64_t *)(prefix + 2u * block);
                for (f = 0; f < frame_count; f++) {
                    hapf_section top, first;
                    uint64_t at;
                    if (results[f] != HapResult_No_Error || !in_dev[f] || TEXTURE_INDEX(f) == 0 || readers[f].view_len < 16u)
                        continue;
                    if (hapf_read_section(readers[f].view, (uint32_t)input_bytes[f], &top) != HapResult_No_Error ||
                        top.type != 0x0Du || top.header_len + 8u > readers[f].view_len ||
                        hapf_read_section(readers[f].view + top.header_len, top.length, &first) != HapResult_No_Error)
                        continue;                                  /* (the planner reports what is wrong with it) */
                    at = (uint64_t)top.header_len + first.header_len + first.length;
                    if (at + 16u <= readers[f].view_len || at >= input_bytes[f] || far_at[f] != at)
                        continue;
                    readers[f].view2_off = at;
                    readers[f].view2 = prefix + block + (size_t)prefix_bytes * f;
                    readers[f].view2_len = input_bytes[f] - at < prefix_bytes ? input_bytes[f] - at : prefix_bytes;
                }
            }

            if (!device_frames || prefix_pass == 1u)
                break;
            for (f = 0; f < frame_count; f++) {
                /* how far the tables of the texture asked for reach: [outer header] section header, instructions header +
                   length (sections: hap.c:137-212) */
                const uint8_t *v;
                size_t have, at = 0;
                hapf_section sec, ins;
                if (results[f] != HapResult_No_Error || !in_dev[f])
                    continue;
                v = readers[f].view;
                have = readers[f].view_len;
                if (TEXTURE_INDEX(f) > 0) {
                    if (!readers[f].view2)
                        continue;
                    v = readers[f].view2;
                    have = readers[f].view2_len;
                } else if (have >= 8u && v[3] == HAP_SECTION_MULTI) {
                    at = (v[0] | v[1] | v[2]) ? 4u : 8u;
                }
                if (have < at + 16u || hapf_read_section(v + at, 0xFFFFFFFFu, &sec) != HapResult_No_Error || (sec.type >> 4) != HAP_NIBBLE_COMPLEX ||
                    hapf_read_section(v + at + sec.header_len, 0xFFFFFFFFu, &ins) != HapResult_No_Error || ins.type != HAP_SECTION_INSTRUCTIONS)
                    continue;
                /* the tables the HOST reads: compressors, sizes, offsets (hap.c:84-88).  The private section's contents
                   are the kernels' business -- an 8K frame's is 800 KB -- only its header is looked at */
                {
                    const size_t ins_at = at + sec.header_len + ins.header_len, ins_end = ins_at + ins.length;
                    size_t q = ins_at, need = 0;
                    unsigned chunks_seen = 0;
                    while (q + 8u <= ins_end) {
                        hapf_section in;
                        if (q + 8u > have) {
                            /* a table header beyond the prefix: behind a compressor table of n entries come 4 n bytes of sizes
                               and perhaps 4 n of offsets */
                            need = q + 16u + 8u * (size_t)chunks_seen + 16u;
                            break;
                        }
                        if (hapf_read_section(v + q, 0xFFFFFFFFu, &in) != HapResult_No_Error)
                            break;
                        if (in.type == HAP_SECTION_COMPRESSORS || in.type == HAP_SECTION_SIZES || in.type == HAP_SECTION_OFFSETS) {
                            if (in.type == HAP_SECTION_COMPRESSORS)
                                chunks_seen = in.length;
                            if (q + in.header_len + in.length > need)
                                need = q + in.header_len + in.length;
                        }
                        q += (size_t)in.header_len + in.length;
                    }
                    if (need > ins_end)
                        need = ins_end;
                    /* (a table that claims to run past the end of its frame is a Bad_Frame for the planner; it does not
                       get to size anybody's scratch) */
                    if (need > input_bytes[f])
                        need = input_bytes[f];
                    if (need > have && need > need_max)
                        need_max = need;
                }
            }
            /* (bounded per frame AND for the batch: one frame's claim decides the prefix of all of them) */
            if (need_max <= prefix_bytes || need_max > ((size_t)4u << 20) ||
                align_up(need_max, 256) * (size_t)frame_count > PREFIX_BATCH_MAX_BYTES)
                break;
            prefix_bytes = align_up(need_max, 256);
            for (f = 0; f < frame_count; f++)
                hapf_reader_free(&readers[f]);
            memset(readers, 0, sizeof(*readers) * frame_count);
            in_stage_bytes = 0;
            memset(in_off, 0, sizeof(size_t) * frame_count);
        }
    }

    HAPB_MARK("prefixes");
    /* 2. plan on the host: sections and tables only (hap_frame.c) */
    for (f = 0; f < frame_count; f++) {
        hapf_texture_plan *p = &plans[f];
        int c;
        if (results[f] != HapResult_No_Error)
            continue;
        hapf_plan_texture(&readers[f], (uint32_t)input_bytes[f], TEXTURE_INDEX(f), 1, p);
        if (output_formats && p->format)
            output_formats[f] = p->format;
        if (p->result != HapResult_No_Error) {
            results[f] = p->result;
            continue;
        }
        if (flags & HAPGPU_DECODE_IGNORE_FRAGMENT_INDEX)
            p->frag_table_offset = 0;
        if (p->mode == HAPGPU_JOB_COMPLEX &&
            !(p->frag_table_offset && p->chunk_count > 0 && p->frag_entries >= (unsigned)p->chunk_count &&
              p->frag_entries % (unsigned)p->chunk_count == 0))
            p->frag_table_offset = 0;
        /* no table, but every chunk as short as one fragment (HAPGPU_ENCODE_FINE_CHUNKS): candidates for the
           block-per-lane decoder, whose starting points a pre-pass can find (snappy_decode_fields.hip) */
        if (p->mode == HAPGPU_JOB_COMPLEX && !p->frag_table_offset && !(flags & HAPGPU_DECODE_NO_FIELD_GUESS) &&
            field_layout_of_format(p->format) && p->chunk_count > 1) {
            unsigned snappy_chunks = 0;
            int fits = 1;
            for (c = 0; c < p->chunk_count; c++) {
                const HapGpuChunkIn *ch = &p->chunks[c];
                if ((ch->codec & 0xFFu) != HAP_NIBBLE_SNAPPY)
                    continue;
                snappy_chunks++;
                if (ch->src_len > HAPGPU_SLOT_DATA_BYTES + 64u + 5u)
                    fits = 0;
            }
            if (fits && snappy_chunks) {
                guess_frame[f] = 1;
                guess_units += snappy_chunks;
            }
        }
    }
    /* (a lane per fragment finds the starting points: worth it from a few thousand fragments on -- one frame's are
       decoded sooner by the generic kernel, a wavefront each) */
    use_guess = guess_units && (guess_units >= 4096u || (flags & HAPGPU_DECODE_GUESS_FIELDS));
    for (f = 0; f < frame_count; f++) {
        hapf_texture_plan *p = &plans[f];
        unsigned units = 0;
        int c;
        /* (the chunks of a frame on that road are not looked over for 64 KiB blocks: one unit each) */
        const int frame_scan = block_scan && !(use_guess && (guess_frame[f] & 1));
        if (results[f] != HapResult_No_Error)
            continue;
        if (p->mode == HAPGPU_JOB_COMPLEX) {
            unsigned per_chunk = 0;
            if (p->frag_table_offset)
                per_chunk = p->frag_entries / (unsigned)p->chunk_count;
            for (c = 0; c < p->chunk_count; c++) {
                HapGpuChunkIn *ch = &p->chunks[c];
                unsigned codec = ch->codec & 0xFFu;
                ch->unit_first = units;
                ch->frag_first = per_chunk * (unsigned)c;
                if (codec == HAP_NIBBLE_SNAPPY)
                    ch->unit_count = per_chunk ? per_chunk
                                               : 1u + (frame_scan ? stream_block_slots(ch->src_len, output_bytes[f]) : 0u);
                else if (codec == HAP_NIBBLE_NONE)
                    ch->unit_count = ch->src_len ? (ch->src_len + COPY_PIECE - 1) / COPY_PIECE : 1u;
                else
                    ch->unit_count = 1u;
                units += ch->unit_count;
                if (codec == HAP_NIBBLE_SNAPPY && !per_chunk) {
                    any_stream = 1;
                    if (ch->src_len > max_stream_src)
                        max_stream_src = ch->src_len;
                }
                if (codec == HAP_NIBBLE_NONE)
                    any_stream = 1;
            }
            if (per_chunk) {
                if (frag_log2_seen && frag_log2_seen != p->frag_log2) {
                    /* one fragment size per launch: later frames fall back to whole-stream units */
                    p->frag_table_offset = 0;
                    units = 0;
                    for (c = 0; c < p->chunk_count; c++) {
                        HapGpuChunkIn *ch = &p->chunks[c];
                        if ((ch->codec & 0xFFu) == HAP_NIBBLE_SNAPPY)
                            ch->unit_count = 1u + (frame_scan ? stream_block_slots(ch->src_len, output_bytes[f]) : 0u);
                        ch->unit_first = units;
                        units += ch->unit_count;
                    }
                    any_stream = 1;
                    max_stream_src = 0xFFFFFFFFu;
                } else {
                    frag_log2_seen = p->frag_log2;
                }
            }
            if (!p->frag_table_offset) {
                /* the 8 KiB blocks of all the frame's streams together: what each could expand to, but no more than the
                   texture holds (+ a ragged last block per stream) -- the slots are handed out on the device */
                unsigned long long fine_sum = 0, streams = 0;
                for (c = 0; c < p->chunk_count; c++)
                    if ((p->chunks[c].codec & 0xFFu) == HAP_NIBBLE_SNAPPY && p->chunks[c].unit_count > 1u) {
                        guess_frame[f] |= 2;
                        scan_chunks += 1u;
                        scan_segs += stream_scan_segments(p->chunks[c].src_len, scan_seg_bytes);
                        scan_words += stream_mark_words(p->chunks[c].src_len, output_bytes[f]);
                        fine_sum += stream_fine_slots(p->chunks[c].src_len, output_bytes[f]);
                        streams += 1u;
                    }
                if (fine_sum > output_bytes[f] / HAPGPU_SCAN_FINE + streams)
                    fine_sum = output_bytes[f] / HAPGPU_SCAN_FINE + streams;
                fine_total += (unsigned)fine_sum;
            }
            total_chunks += (unsigned)p->chunk_count;
            if ((unsigned)p->chunk_count > max_chunks)
                max_chunks = (unsigned)p->chunk_count;
        } else if (p->mode == HAPGPU_JOB_RAW) {
            units = p->section_length ? (p->section_length + COPY_PIECE - 1) / COPY_PIECE : 1u;
            any_stream = 1;
        } else {
            if (p->section_length > max_stream_src)
                max_stream_src = p->section_length;
            units = 1u + (block_scan ? stream_block_slots(p->section_length, output_bytes[f]) : 0u);
            if (units > 1u) {
                guess_frame[f] |= 2;
                scan_chunks += 1u;
                scan_segs += stream_scan_segments(p->section_length, scan_seg_bytes);
                scan_words += stream_mark_words(p->section_length, output_bytes[f]);
                fine_total += stream_fine_slots(p->section_length, output_bytes[f]);
            }
            any_stream = 1;
        }
        job_of_frame[f] = live++;
        total_units += units;
        p->frag_entries = p->frag_table_offset ? p->frag_entries : 0;
        p->unit_count = units;
        if (!out_dev[f]) {
            out_off[f] = out_stage_bytes + 1;
            out_stage_bytes += align_up(output_bytes[f], 256);
        }
    }
    if (live == 0)
        goto finish;
    /* streams without a table that the block scan cuts into 8 KiB pieces (plain hap.h frames of this library: one or a
       few chunks of many fragments): the starting points inside every piece the scan lists can be had from the scan's own
       records (r06: a wavefront per piece looks up the windows that hold every ceil(N / 64)-th element and walks a dozen
       elements each; until r05 a lane per piece walked its ~400 elements twice: 1.95 ms per 60 8K frames, 0.87 now), and
       the block-per-lane decoder then takes the pieces that are field-stream fragments; the others (another encoder's)
       stay with the generic kernel.  Calls, plain 8K frames: 1 frame 0.435 ms against 0.412 through the generic kernel
       alone, 8 frames 0.84 against 1.09, 60 frames 4.03 against 6.30 (r05, lane per piece: 4.98): from two frames' pieces on. */
    use_scan_guess = fine_total && !(flags & HAPGPU_DECODE_NO_FIELD_GUESS);

    /* 3. device descriptors */
    hjobs = (HapGpuDecodeJob *)hapgpu_rt_pinned_scratch(rt, P_JOBS, sizeof(HapGpuDecodeJob) * live);
    hchunks = (HapGpuChunkIn *)hapgpu_rt_pinned_scratch(rt, P_CHUNKS, sizeof(HapGpuChunkIn) * (total_chunks + 1u));
    djobs = (HapGpuDecodeJob *)hapgpu_rt_device_scratch(rt, D_JOBS, sizeof(HapGpuDecodeJob) * live);
    dchunks = (HapGpuChunkIn *)hapgpu_rt_device_scratch(rt, D_CHUNKS, sizeof(HapGpuChunkIn) * (total_chunks + 1u));
    /* (the fine block units of the block scan live behind the ordinary ones: [total_units, total_units + fine_total)) */
    dunits = (HapGpuDecodeUnit *)hapgpu_rt_device_scratch(rt, D_UNITS, sizeof(HapGpuDecodeUnit) * ((size_t)total_units + fine_total + 1u));
    if (in_stage_bytes)
        in_stage = (uint8_t *)hapgpu_rt_device_scratch(rt, D_IN_STAGE, in_stage_bytes);
    if (out_stage_bytes)
        out_stage = (uint8_t *)hapgpu_rt_device_scratch(rt, D_OUT_STAGE, out_stage_bytes);
    if (!hjobs || !hchunks || !djobs || !dchunks || !dunits || (in_stage_bytes && !in_stage) ||
        (out_stage_bytes && !out_stage)) {
        rc = 1;
        goto fail_alloc;
    }
    if (scan_chunks) {
        /* one arena: chunk table | segment summaries | block positions | fine work list | joins | window records */
        const size_t o_segs = align_up(sizeof(HapGpuScanChunk) * scan_chunks, 64);
        const size_t o_bpos = o_segs + sizeof(HapGpuScanSegment) * scan_segs;
        const size_t o_work = align_up(o_bpos + sizeof(uint32_t) * scan_words, 64);
        const size_t o_joins = align_up(o_work + sizeof(uint32_t) * ((size_t)fine_total + 2u), 64);   /* count | list | pool cursor */
        const size_t o_recs = align_up(o_joins + (size_t)16u * scan_segs, 64);
        uint8_t *arena = (uint8_t *)hapgpu_rt_device_scratch(rt, D_SCAN, o_recs + (size_t)512u * scan_segs);
        hscan = (HapGpuScanChunk *)hapgpu_rt_pinned_scratch(rt, P_SCAN, sizeof(HapGpuScanChunk) * scan_chunks);
        if (!arena || !hscan) {
            rc = 1;
            goto fail_alloc;
        }
        dscan = (HapGpuScanChunk *)arena;
        dsegs = (HapGpuScanSegment *)(arena + o_segs);
        dbpos = (uint32_t *)(arena + o_bpos);
        dwork = (uint32_t *)(arena + o_work);
        djoins = arena + o_joins;
        drecs = arena + o_recs;
    }
    if (use_guess || use_scan_guess) {
        /* (a table per unit slot; the scan's pieces are the slots behind the ordinary ones) */
        dguess = (uint8_t *)hapgpu_rt_device_scratch(rt, D_GUESS, (size_t)HAP_GROUP_TABLE_BYTES *
                                                     ((size_t)total_units + (use_scan_guess ? fine_total : 0u) + 1u));
        if (!dguess) {
            rc = 1;
            goto fail_alloc;
        }
    }
    {
        unsigned chunk_cursor = 0, unit_cursor = 0, scan_cursor = 0, seg_cursor = 0, word_cursor = 0, fine_cursor = 0;
        request_marks marks;
        const void *staged_src = NULL;
        unsigned long staged_len = 0;
        const uint8_t *staged_dev = NULL;
        marks.count = 0;
        marks.requested = NULL;
        for (f = 0; f < frame_count; f++) {
            hapf_texture_plan *p = &plans[f];
            HapGpuDecodeJob *job;
            const uint8_t *frame_dev;
            unsigned units;
            if (results[f] != HapResult_No_Error)
                continue;
            units = p->unit_count;
            job = &hjobs[job_of_frame[f]];
            memset(job, 0, sizeof(*job));
            if (in_off[f] && inputs[f] == staged_src && input_bytes[f] == staged_len) {
                /* (the entry before this one was the same host frame -- HapGpuDecodeFramesRGBA asks for both textures of a
                   frame in one call: one upload serves both) */
                frame_dev = staged_dev;
            } else if (in_off[f]) {
                uint8_t *d = in_stage + (in_off[f] - 1);
                rc |= hapgpu_rt_h2d(rt, d, inputs[f], input_bytes[f]);
                frame_dev = d;
                staged_src = inputs[f];
                staged_len = input_bytes[f];
                staged_dev = d;
            } else {
                frame_dev = (const uint8_t *)inputs[f];
            }
            job->dst = (uint64_t)(uintptr_t)(out_off[f] ? out_stage + (out_off[f] - 1) : (uint8_t *)outputs[f]);
            job->dst_cap = output_bytes[f];
            job->mode = p->mode;
            job->unit_count = units;
            job->units = (uint64_t)(uintptr_t)(dunits + unit_cursor);
            job->status = HapResult_Internal_Error;
            if (p->mode == HAPGPU_JOB_COMPLEX) {
                job->payload = (uint64_t)(uintptr_t)(frame_dev + p->payload_offset);
                job->payload_len = p->payload_length;
                job->chunk_count = (uint32_t)p->chunk_count;
                job->chunks = (uint64_t)(uintptr_t)(dchunks + chunk_cursor);
                if (p->frag_table_offset) {
                    job->frag_sizes = (uint64_t)(uintptr_t)(frame_dev + p->frag_table_offset);
                    job->frag_log2 = p->frag_log2;
                    job->frag_entries = p->frag_entries;
                    job->reserved = p->frag_gran_log2 | (p->frag_window256 << 8);
                    /* bits 0..2: plain fragment kernels needed, bits 4..6: windowed ones (8 KiB fragments whose
                       table promises offsets of at most 3 KiB), bits 8 / 9: field streams (table version 3) */
                    if (p->frag_tiles_offset && p->frag_fields && !(flags & HAPGPU_DECODE_IGNORE_HALF_TILES)) {
                        job->fields_period = p->frag_fields;
                        job->group_tables = (uint64_t)(uintptr_t)(frame_dev + p->frag_tiles_offset);
                        frag_kinds |= p->frag_fields == 4u ? 0x100u : p->frag_fields == 2u ? 0x200u : p->frag_fields == 8u ? 0x800u : 0x400u;
                    } else if (p->frag_log2 == 13u && p->frag_window256 != 0 && p->frag_window256 <= HAP_FRAGMENT_WINDOW_256)
                        frag_kinds |= 16u << p->frag_gran_log2;
                    else
                        frag_kinds |= 1u << p->frag_gran_log2;
                }
                if (dguess && use_guess && (guess_frame[f] & 1)) {
                    const unsigned layout = field_layout_of_format(p->format);
                    job->fields_period = layout;
                    job->group_tables = (uint64_t)(uintptr_t)dguess;
                    job->reserved |= 1u << 16;
                    frag_kinds |= layout == 4u ? 0x100u : layout == 2u ? 0x200u : layout == 8u ? 0x800u : 0x400u;
                }
                if (p->chunk_count > 0)
                    memcpy(hchunks + chunk_cursor, p->chunks, sizeof(HapGpuChunkIn) * (size_t)p->chunk_count);
                chunk_cursor += (unsigned)p->chunk_count;
                if (scan_chunks && !p->frag_table_offset) {
                    int c;
                    for (c = 0; c < p->chunk_count; c++) {
                        const HapGpuChunkIn *ch = &p->chunks[c];
                        if ((ch->codec & 0xFFu) != HAP_NIBBLE_SNAPPY || ch->unit_count <= 1u)
                            continue;
                        scan_entry(&hscan[scan_cursor++], unit_cursor + ch->unit_first, ch->src_len, output_bytes[f],
                                   dbpos, &seg_cursor, &word_cursor, scan_seg_bytes, &fine_cursor);
                    }
                }
            } else {
                job->payload = (uint64_t)(uintptr_t)(frame_dev + p->section_offset);
                job->payload_len = p->section_length;
                if (scan_chunks && p->mode == HAPGPU_JOB_SNAPPY && units > 1u)
                    scan_entry(&hscan[scan_cursor++], unit_cursor, p->section_length, output_bytes[f], dbpos, &seg_cursor,
                               &word_cursor, scan_seg_bytes, &fine_cursor);
            }
            if (dguess && use_scan_guess && (guess_frame[f] & 2) && field_layout_of_format(p->format) && !job->fields_period) {
                const unsigned layout = field_layout_of_format(p->format);
                job->fields_period = layout;
                job->group_tables = (uint64_t)(uintptr_t)dguess;
                job->reserved |= 1u << 16;
                /* 0x1000: units for the block-per-lane decoder may appear among the scan's pieces as well */
                frag_kinds |= (layout == 4u ? 0x100u : layout == 2u ? 0x200u : 0x400u) | 0x1000u;
            }
            unit_cursor += units;
        }
        HAPB_MARK("host plan");
        rc |= hapgpu_rt_h2d(rt, djobs, hjobs, sizeof(HapGpuDecodeJob) * live);
        if (total_chunks)
            rc |= hapgpu_rt_h2d(rt, dchunks, hchunks, sizeof(HapGpuChunkIn) * total_chunks);
        rc |= hapgpu_k_decode_plan(rt, djobs, live, dunits, total_units, frag_log2_seen ? max_chunks : 0u);

        /* hap.h callback contract (single-frame HapDecode only): the client is asked to "run" the
           chunks once planning succeeded and there is more than one (reference hap.c:852-862) */
        if (callback && frame_count == 1 && !rc && results[0] == HapResult_No_Error &&
            plans[0].mode == HAPGPU_JOB_COMPLEX && plans[0].chunk_count > 1) {
            unsigned char *req;
            int c, all = 1;
            rc |= hapgpu_rt_d2h(rt, hjobs, djobs, sizeof(HapGpuDecodeJob));
            rc |= hapgpu_rt_sync(rt);
            if (!rc && hjobs[0].status == HapResult_No_Error) {
                req = (unsigned char *)calloc((size_t)plans[0].chunk_count, 1);
                if (!req) {
                    rc = 1;
                } else {
                    marks.count = (unsigned)plans[0].chunk_count;
                    marks.requested = req;
                    if (ctx->preset_marks && ctx->preset_count == marks.count)
                        memcpy(req, ctx->preset_marks, marks.count);      /* retry: the client was asked already */
                    else
                        callback(mark_chunk, &marks, marks.count, callback_info);
                    free(client_marks);
                    client_marks = (unsigned char *)malloc(marks.count);
                    if (client_marks) {
                        memcpy(client_marks, req, marks.count);
                        client_marks_count = marks.count;
                    }
                    for (c = 0; c < plans[0].chunk_count; c++)
                        if (!req[c])
                            all = 0;
                    if (!all) {
                        /* chunks the client never asked for stay undecoded: blank their units
                           (HAPGPU_UNIT_SKIP is 0; unit slots of consecutive chunks are contiguous) */
                        /* a staged (host) output must come back unchanged where nothing is decoded */
                        if (out_off[0])
                            rc |= hapgpu_rt_h2d(rt, out_stage + (out_off[0] - 1), outputs[0],
                                                output_bytes[0] < hjobs[0].bytes_used ? output_bytes[0] : (size_t)hjobs[0].bytes_used);
                        for (c = 0; c < plans[0].chunk_count;) {
                            unsigned first, units = 0;
                            if (req[c]) {
                                c++;
                                continue;
                            }
                            first = plans[0].chunks[c].unit_first;
                            while (c < plans[0].chunk_count && !req[c] &&
                                   plans[0].chunks[c].unit_first == first + units) {
                                units += plans[0].chunks[c].unit_count;
                                c++;
                            }
                            rc |= hapgpu_rt_zero(rt, dunits + first, sizeof(HapGpuDecodeUnit) * (size_t)units);
                        }
                    }
                    free(req);
                }
            }
        }
        /* streams of other encoders (no fragment table): find their independent 64 KiB blocks first */
        if (scan_chunks && scan_cursor == scan_chunks) {
            rc |= hapgpu_rt_h2d(rt, dscan, hscan, sizeof(HapGpuScanChunk) * scan_chunks);
            rc |= hapgpu_rt_zero(rt, dwork, sizeof(uint32_t));
            if (fine_total)
                rc |= hapgpu_rt_zero(rt, dwork + 1u + fine_total, sizeof(uint32_t));
            if (frag_kinds & 0x1000u)       /* (the decoder below walks all the pieces' slots, not only the listed ones) */
                rc |= hapgpu_rt_zero(rt, dunits + total_units, sizeof(HapGpuDecodeUnit) * (size_t)fine_total);
            rc |= hapgpu_k_scan_blocks(rt, dunits, djobs, dscan, scan_chunks, dsegs, drecs, djoins, scan_segs,
                                       fine_total ? dwork : NULL, total_units, fine_total);
        }
        if (dguess && use_guess)
            rc |= hapgpu_k_guess_group_tables(rt, dunits, total_units, djobs, NULL, 0u);
        if (dguess && (frag_kinds & 0x1000u)) {
            if (scan_chunks && scan_cursor == scan_chunks)
                rc |= hapgpu_k_guess_group_tables(rt, dunits, total_units + fine_total, djobs, dwork, fine_total);
            else
                frag_kinds &= ~0x1000u;
        }
        rc |= hapgpu_k_snappy_decode(rt, dunits, total_units, djobs, frag_log2_seen, frag_kinds,
                                     /* 3: every stream is as short as one 8 KiB fragment (frames written with
                                        HAPGPU_ENCODE_FINE_CHUNKS): the 2 KiB ring of the block-scan launches instead of the 32 KiB one, 30 wavefronts per CU instead of 4 */
                                     any_stream ? (scan_chunks ? 2 : (max_stream_src <= HAPGPU_SLOT_DATA_BYTES + 64u ? 3 : 1)) : 0,
                                     (scan_chunks && scan_cursor == scan_chunks && fine_total) ? dwork : NULL, fine_total);
        rc |= hapgpu_rt_d2h(rt, hjobs, djobs, sizeof(HapGpuDecodeJob) * live);
        HAPB_MARK("launched");
        rc |= hapgpu_rt_sync(rt);
        HAPB_MARK("completed");
        if (rc)
            goto fail_alloc;
    }

    /* 4. results */
    {
        int copied = 0;
        for (f = 0; f < frame_count; f++) {
            HapGpuDecodeJob *job;
            if (results[f] != HapResult_No_Error)
                continue;
            job = &hjobs[job_of_frame[f]];
            if (job->status == HAPGPU_STATUS_INDEX_MISMATCH) {
                need_retry = 1;
                results[f] = HAPGPU_STATUS_INDEX_MISMATCH;
                continue;
            }
            results[f] = job->status;
            if (job->status == HapResult_No_Error) {
                if (output_used)
                    output_used[f] = (unsigned long)job->bytes_used;
                if (out_off[f]) {
                    rc |= hapgpu_rt_d2h(rt, outputs[f], out_stage + (out_off[f] - 1), (size_t)job->bytes_used);
                    copied = 1;
                }
            }
        }
        if (copied)
            rc |= hapgpu_rt_sync(rt);
        if (rc)
            goto fail_alloc;
    }

    /* 5. frames whose fragment table did not describe their streams: decode them the generic way */
    if (need_retry) {
        for (f = 0; f < frame_count; f++) {
            if (results[f] != HAPGPU_STATUS_INDEX_MISMATCH)
                continue;
            results[f] = HapResult_No_Error;
            ctx->table_fallbacks += 1;
            if (frame_count == 1 && client_marks) {
                ctx->preset_marks = client_marks;
                ctx->preset_count = client_marks_count;
            }
            hapb_decode(ctx, 1, &inputs[f], &input_bytes[f], TEXTURE_INDEX(f), &outputs[f], &output_bytes[f],
                        output_used ? &output_used[f] : NULL, output_formats ? &output_formats[f] : NULL,
                        &results[f], (flags | HAPGPU_DECODE_IGNORE_FRAGMENT_INDEX | HAPGPU_DECODE_NO_BLOCK_SCAN | HAPGPU_DECODE_NO_FIELD_GUESS) &
                                     ~HAPGPU_DECODE_GUESS_FIELDS,
                        frame_count == 1 ? callback : NULL, callback_info);
            ctx->preset_marks = NULL;
            ctx->preset_count = 0;
        }
    }

finish:
    HAPB_MARK("results");
    for (f = 0; f < frame_count; f++) {
        if (results[f] != HapResult_No_Error && first_error == HapResult_No_Error)
            first_error = results[f];
        hapf_plan_free(&plans[f]);
        hapf_reader_free(&readers[f]);
    }
    free(plans); free(readers); free(fetchers); free(in_off); free(out_off); free(job_of_frame);
    free(in_dev); free(out_dev); free(client_marks); free(guess_frame);
    return first_error;

fail_alloc:
    for (f = 0; f < frame_count; f++) {
        if (results[f] == HapResult_No_Error || results[f] == HAPGPU_STATUS_INDEX_MISMATCH)
            results[f] = HapResult_Internal_Error;
        if (plans) hapf_plan_free(&plans[f]);
        if (readers) hapf_reader_free(&readers[f]);
    }
    free(plans); free(readers); free(fetchers); free(in_off); free(out_off); free(job_of_frame);
    free(in_dev); free(out_dev); free(client_marks); free(guess_frame);
    return HapResult_Internal_Error;
}

/* ============================================================= frames to pixels */
/* Hap frames -> RGBA8 pictures with nothing but frames going in and pixels coming out: the second stage of a slice of
   frames is undone into a scratch of block textures (one batch through hapb_decode: texture 0, and the RGTC1 alpha
   plane of Hap Q Alpha frames), then every picture is expanded by the block decoder on the same stream.  What a
   player without texture sampling of its own asks of a Hap decoder (the reference leaves this step to the GPU's
   texture units, hap.h:92-95 "the texture format the frame decodes to"). */
#define RGBA_SLICE_BYTES ((size_t)4u << 30)     /* block textures held at a time */
unsigned hapb_decode_rgba(HapGpuContext *ctx, unsigned frame_count, const void *const *inputs,
                          const unsigned long *input_bytes, unsigned texture_count, void *const *rgba_frames,
                          unsigned width, unsigned height, unsigned long row_bytes, unsigned *results, unsigned flags)
{
    hapgpu_rt *rt = ctx->rt;
    size_t blocks, per_frame, alpha_off, rgba_bytes, slice, done;
    unsigned first_error = HapResult_No_Error, f;
    const void **in;
    unsigned long *in_bytes, *caps, *used;
    void **outs;
    unsigned *idx, *fmts, *res;
    if (frame_count == 0)
        return HapResult_No_Error;
    if (!results)
        return HapResult_Bad_Arguments;
    if (context_busy(ctx, results, frame_count))
        return HapResult_Internal_Error;
    if (!inputs || !input_bytes || !rgba_frames || texture_count == 0 || texture_count > 2 || width == 0 ||
        height == 0 || (width & 3u) || (height & 3u) || row_bytes < (unsigned long)width * 4ul || (row_bytes & 15u)) {
        for (f = 0; f < frame_count; f++)
            results[f] = HapResult_Bad_Arguments;
        return HapResult_Bad_Arguments;
    }
    blocks = (size_t)(width / 4u) * (height / 4u);
    alpha_off = align_up(blocks * 16u, 256);
    per_frame = alpha_off + (texture_count == 2 ? align_up(blocks * 8u, 256) : 0u);
    rgba_bytes = (size_t)row_bytes * (height - 1u) + (size_t)width * 4u;
    slice = RGBA_SLICE_BYTES / per_frame;
    if (slice == 0)
        slice = 1;
    if (slice > frame_count)
        slice = frame_count;
    if (slice * texture_count > 32768u)
        slice = 32768u / texture_count;
    in = (const void **)malloc(sizeof(*in) * slice * texture_count);
    in_bytes = (unsigned long *)malloc(sizeof(*in_bytes) * slice * texture_count * 3u);
    outs = (void **)malloc(sizeof(*outs) * slice * texture_count);
    idx = (unsigned *)malloc(sizeof(*idx) * slice * texture_count * 3u);
    if (!in || !in_bytes || !outs || !idx) {
        free(in); free(in_bytes); free(outs); free(idx);
        for (f = 0; f < frame_count; f++)
            results[f] = HapResult_Internal_Error;
        return HapResult_Internal_Error;
    }
    caps = in_bytes + slice * texture_count;
    used = caps + slice * texture_count;
    fmts = idx + slice * texture_count;
    res = fmts + slice * texture_count;
    for (done = 0; done < frame_count; done += slice) {
        const unsigned n = (unsigned)(frame_count - done < slice ? frame_count - done : slice);
        uint8_t *textures = (uint8_t *)hapgpu_rt_device_scratch(rt, D_BC_TEX, per_frame * n);
        uint8_t *stage = NULL;
        unsigned t;
        int rc = 0;
        if (!textures) {
            for (f = 0; f < n; f++)
                results[done + f] = HapResult_Internal_Error;
            first_error = first_error ? first_error : HapResult_Internal_Error;
            continue;
        }
        for (f = 0; f < n; f++)
            for (t = 0; t < texture_count; t++) {
                const size_t e = (size_t)f * texture_count + t;
                in[e] = inputs[done + f];
                in_bytes[e] = input_bytes[done + f];
                outs[e] = textures + per_frame * f + (t ? alpha_off : 0u);
                caps[e] = (unsigned long)(t ? blocks * 8u : blocks * 16u);
                idx[e] = t;
                used[e] = 0;
                fmts[e] = 0;
            }
        ctx->decode_indices = idx;
        hapb_decode(ctx, n * texture_count, in, in_bytes, 0, outs, caps, used, fmts, res, flags, NULL, NULL);
        {
            /* one block-decode launch per texture format present in the slice: [textures][alpha planes][pictures] in a
               small device table, pictures of other formats (or that failed) with a texture address of 0 */
            uint64_t *htab = (uint64_t *)hapgpu_rt_pinned_scratch(rt, P_BC_PTRS, sizeof(uint64_t) * 9u * n);
            uint64_t *dtab = (uint64_t *)hapgpu_rt_device_scratch(rt, D_BC_PTRS, sizeof(uint64_t) * 9u * n);
            static const unsigned kinds[3] = {HapTextureFormat_RGB_DXT1, HapTextureFormat_RGBA_DXT5, HapTextureFormat_YCoCg_DXT5};
            unsigned present = 0, k;
            if (!htab || !dtab) {
                for (f = 0; f < n; f++)
                    results[done + f] = HapResult_Internal_Error;
                first_error = first_error ? first_error : HapResult_Internal_Error;
                continue;
            }
            memset(htab, 0, sizeof(uint64_t) * 9u * n);
            for (f = 0; f < n; f++) {
                const size_t e = (size_t)f * texture_count;
                const unsigned fmt = fmts[e];
                void *dst = rgba_frames[done + f];
                unsigned r = res[e];
                if (r == HapResult_No_Error && texture_count == 2)
                    r = res[e + 1];
                if (r == HapResult_No_Error && !dst)
                    r = HapResult_Bad_Arguments;
                /* the frame must hold what the caller's geometry says: a colour texture the block decoder knows, of
                   exactly width x height, and (two textures) an RGTC1 plane of the same geometry */
                if (r == HapResult_No_Error &&
                    ((fmt != HapTextureFormat_RGB_DXT1 && fmt != HapTextureFormat_RGBA_DXT5 && fmt != HapTextureFormat_YCoCg_DXT5) ||
                     used[e] != blocks * (fmt == HapTextureFormat_RGB_DXT1 ? 8u : 16u) ||
                     (texture_count == 2 && (fmts[e + 1] != HapTextureFormat_A_RGTC1 || used[e + 1] != blocks * 8u))))
                    r = HapResult_Bad_Arguments;
                if (r == HapResult_No_Error && !is_dev(ctx, dst)) {
                    if (!stage)
                        stage = (uint8_t *)hapgpu_rt_device_scratch(rt, D_RGBA_STAGE, align_up(rgba_bytes, 256) * n);
                    dst = stage ? stage + align_up(rgba_bytes, 256) * f : NULL;
                    if (!dst)
                        r = HapResult_Internal_Error;
                } else if (r == HapResult_No_Error && ((uintptr_t)dst & 15u)) {
                    r = HapResult_Bad_Arguments;
                }
                if (r == HapResult_No_Error) {
                    k = fmt == HapTextureFormat_RGB_DXT1 ? 0u : fmt == HapTextureFormat_RGBA_DXT5 ? 1u : 2u;
                    present |= 1u << k;
                    htab[(size_t)k * 3u * n + f] = (uint64_t)(uintptr_t)outs[e];
                    htab[(size_t)k * 3u * n + n + f] = texture_count == 2 ? (uint64_t)(uintptr_t)outs[e + 1] : 0u;
                    htab[(size_t)k * 3u * n + 2u * n + f] = (uint64_t)(uintptr_t)dst;
                }
                results[done + f] = r;
            }
            if (present) {
                rc |= hapgpu_rt_h2d(rt, dtab, htab, sizeof(uint64_t) * 9u * n);
                for (k = 0; k < 3u; k++)
                    if (present & (1u << k))
                        rc |= hapgpu_k_block_decode_batch(rt, dtab + (size_t)k * 3u * n, n, texture_count == 2, width, height, kinds[k],
                                                          row_bytes);
                for (f = 0; f < n; f++)
                    if (results[done + f] == HapResult_No_Error && stage && !is_dev(ctx, rgba_frames[done + f])) {
                        /* (row by row when the client's rows are longer than the picture's: what lies between them -- the
                           rest of a larger image, perhaps -- is not this call's to overwrite) */
                        if (row_bytes == (unsigned long)width * 4ul)
                            rc |= hapgpu_rt_d2h(rt, rgba_frames[done + f], stage + align_up(rgba_bytes, 256) * f, rgba_bytes);
                        else
                            rc |= hapgpu_rt_d2h_rows(rt, rgba_frames[done + f], row_bytes, stage + align_up(rgba_bytes, 256) * f, row_bytes,
                                                     (size_t)width * 4u, height);
                    }
            }
        }
        rc |= hapgpu_rt_sync(rt);
        if (rc)
            for (f = 0; f < n; f++)
                if (results[done + f] == HapResult_No_Error)
                    results[done + f] = HapResult_Internal_Error;
        for (f = 0; f < n; f++)
            if (results[done + f] != HapResult_No_Error && first_error == HapResult_No_Error)
                first_error = results[done + f];
    }
    free(in); free(in_bytes); free(outs); free(idx);
    return first_error;
}

/* ============================================================= join on the device */
/* Sink of hapj_join for frames in device memory: header bytes are collected on the host and uploaded in one copy, the
   groups' tables and payloads become device-to-device moves of the gather kernel (pieces of at most 64 KiB, one
   wavefront each).  Two launches: what the host made first, then the moves (table contents land on top of the zeros
   of their header region). */
#define JOIN_PIECE 65536u
typedef struct move_list {
    HapGpuCopyEntry *e;
    size_t count, cap;
} move_list;

typedef struct device_sink {
    uint8_t *bytes;               /* host-made bytes, 16-byte aligned pieces */
    size_t bytes_used, bytes_cap;
    move_list puts, moves;        /* puts: src = offset into `bytes` until the upload address is known */
    const void *const *frames;
    uint8_t *out;
} device_sink;

static int move_append(move_list *l, uint64_t src, uint64_t dst, size_t len)
{
    while (len) {
        const size_t n = len < JOIN_PIECE ? len : JOIN_PIECE;
        if (l->count == l->cap) {
            const size_t cap = l->cap ? 2u * l->cap : 256u;
            HapGpuCopyEntry *e = (HapGpuCopyEntry *)realloc(l->e, cap * sizeof(*e));
            if (!e)
                return 1;
            l->e = e;
            l->cap = cap;
        }
        l->e[l->count].src = src;
        l->e[l->count].dst = dst;
        l->e[l->count].len = (uint32_t)n;
        l->e[l->count].reserved = 0;
        l->count++;
        src += n; dst += n; len -= n;
    }
    return 0;
}

static int device_put(void *user, uint64_t dst_off, const void *src, size_t len)
{
    device_sink *d = (device_sink *)user;
    const size_t padded = (len + 15u) & ~(size_t)15u;
    if (len == 0)
        return 0;
    if (d->bytes_used + padded > d->bytes_cap) {
        const size_t cap = 2u * (d->bytes_used + padded) + 4096u;
        uint8_t *b = (uint8_t *)realloc(d->bytes, cap);
        if (!b)
            return 1;
        d->bytes = b;
        d->bytes_cap = cap;
    }
    memcpy(d->bytes + d->bytes_used, src, len);
    memset(d->bytes + d->bytes_used + len, 0, padded - len);
    if (move_append(&d->puts, (uint64_t)d->bytes_used, (uint64_t)(uintptr_t)(d->out + dst_off), len))
        return 1;
    d->bytes_used += padded;
    return 0;
}

static int device_move(void *user, unsigned group, uint64_t src_off, uint64_t dst_off, size_t len)
{
    device_sink *d = (device_sink *)user;
    return move_append(&d->moves, (uint64_t)(uintptr_t)((const uint8_t *)d->frames[group] + src_off),
                       (uint64_t)(uintptr_t)(d->out + dst_off), len);
}

unsigned hapb_join_device(HapGpuContext *ctx, unsigned group_count, const void *const *frames,
                          const unsigned long *frame_bytes, void *output, unsigned long output_bytes,
                          unsigned long *output_used)
{
    hapgpu_rt *rt = ctx->rt;
    hapf_reader *readers;
    fetch_ctx *fetchers;
    uint8_t *prefix, *dprefix;
    uint64_t *hptr, *dptr;
    device_sink ds;
    hapj_sink sink;
    unsigned g, result, rc = 0;

    if (group_count == 0 || !frames || !frame_bytes || !output || !output_used)
        return HapResult_Bad_Arguments;
    if (context_busy(ctx, NULL, 0))
        return HapResult_Internal_Error;
    for (g = 0; g < group_count; g++)
        if (!frames[g] || frame_bytes[g] > 0xFFFFFFFFul || !is_dev(ctx, frames[g]))
            return HapResult_Bad_Arguments;
    if (!is_dev(ctx, output))
        return HapResult_Bad_Arguments;
    readers = (hapf_reader *)calloc(group_count, sizeof(*readers));
    fetchers = (fetch_ctx *)calloc(group_count, sizeof(*fetchers));
    hptr = (uint64_t *)hapgpu_rt_pinned_scratch(rt, P_PTRS, sizeof(uint64_t) * 2u * group_count);
    dptr = (uint64_t *)hapgpu_rt_device_scratch(rt, D_PTRS, sizeof(uint64_t) * 2u * group_count);
    dprefix = (uint8_t *)hapgpu_rt_device_scratch(rt, D_PREFIX, (size_t)PREFIX_BYTES * group_count);
    prefix = (uint8_t *)hapgpu_rt_pinned_scratch(rt, P_PREFIX, (size_t)PREFIX_BYTES * group_count);
    if (!readers || !fetchers || !hptr || !dptr || !dprefix || !prefix) {
        free(readers); free(fetchers);
        return HapResult_Internal_Error;
    }
    /* the groups' headers come to the host (one gather + one copy); what the planner needs beyond the prefix -- the
       tables of frames with many chunks -- it fetches on demand */
    for (g = 0; g < group_count; g++) {
        hptr[g] = (uint64_t)(uintptr_t)frames[g];
        hptr[group_count + g] = frame_bytes[g];
    }
    rc |= (unsigned)hapgpu_rt_h2d(rt, dptr, hptr, sizeof(uint64_t) * 2u * group_count);
    rc |= (unsigned)hapgpu_k_gather_prefixes(rt, dptr, dptr + group_count, group_count, PREFIX_BYTES, dprefix);
    rc |= (unsigned)hapgpu_rt_d2h(rt, prefix, dprefix, (size_t)PREFIX_BYTES * group_count);
    rc |= (unsigned)hapgpu_rt_sync(rt);
    if (rc) {
        free(readers); free(fetchers);
        return HapResult_Internal_Error;
    }
    for (g = 0; g < group_count; g++) {
        const size_t n = frame_bytes[g] < PREFIX_BYTES ? frame_bytes[g] : PREFIX_BYTES;
        hapf_reader_init_host(&readers[g], prefix + (size_t)PREFIX_BYTES * g, n);
        fetchers[g].ctx = ctx;
        fetchers[g].device_frame = (const uint8_t *)frames[g];
        readers[g].fetch = fetch_from_device;
        readers[g].user = &fetchers[g];
        readers[g].total_len = frame_bytes[g];
    }
    memset(&ds, 0, sizeof(ds));
    ds.frames = frames;
    ds.out = (uint8_t *)output;
    sink.user = &ds;
    sink.put = device_put;
    sink.move = device_move;
    result = hapj_join(group_count, readers, frame_bytes, &sink, output_bytes, output_used);
    if (result == HapResult_No_Error) {
        const size_t entries = ds.puts.count + ds.moves.count;
        uint8_t *dbytes = (uint8_t *)hapgpu_rt_device_scratch(rt, D_JOBS, ds.bytes_used + 16u);
        HapGpuCopyEntry *dmoves = (HapGpuCopyEntry *)hapgpu_rt_device_scratch(rt, D_COPIES, sizeof(HapGpuCopyEntry) * (entries + 1u));
        size_t i;
        if (!dbytes || !dmoves) {
            result = HapResult_Internal_Error;
        } else {
            for (i = 0; i < ds.puts.count; i++)
                ds.puts.e[i].src += (uint64_t)(uintptr_t)dbytes;
            if (ds.bytes_used)
                rc |= (unsigned)hapgpu_rt_h2d(rt, dbytes, ds.bytes, ds.bytes_used);
            if (ds.puts.count)
                rc |= (unsigned)hapgpu_rt_h2d(rt, dmoves, ds.puts.e, sizeof(HapGpuCopyEntry) * ds.puts.count);
            if (ds.moves.count)
                rc |= (unsigned)hapgpu_rt_h2d(rt, dmoves + ds.puts.count, ds.moves.e, sizeof(HapGpuCopyEntry) * ds.moves.count);
            if (ds.puts.count)
                rc |= (unsigned)hapgpu_k_frame_gather(rt, dmoves, (unsigned)ds.puts.count);
            if (ds.moves.count)
                rc |= (unsigned)hapgpu_k_frame_gather(rt, dmoves + ds.puts.count, (unsigned)ds.moves.count);
            rc |= (unsigned)hapgpu_rt_sync(rt);     /* (the lists live in pageable memory: nothing may still read them) */
            if (rc)
                result = HapResult_Internal_Error;
        }
    }
    free(ds.bytes); free(ds.puts.e); free(ds.moves.e);
    for (g = 0; g < group_count; g++)
        hapf_reader_free(&readers[g]);
    free(readers); free(fetchers);
    return result;
}
