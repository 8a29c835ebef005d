/*
 * hap_devices.c -- one batch over several contexts (devices) from C: SURVEY.md 8(e), BASELINE.json north_star
 * ("sharded across the 8 GPUs of one node by assigning independent frames per GPU").
 *
 * Frames are independent, so the multi-GPU form of the batched calls needs no collective: frame f goes to context
 * f mod N, every context works on its share on its own stream from a host thread of its own, and the per-frame
 * results land in the caller's arrays at the frame's own index.  It replaces what a reference client does with a pool
 * of threads each calling HapEncode / HapDecode (hap.h:98-140; the chunk fan-out of hap.c:852-862 has no per-device
 * notion at all).  Contexts may share a device (tests play N = 2, 3, 8 on one GPU); the bytes are the same as those of
 * the single-context call whatever N is.
 */
#include "hap_batch.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct share {
    HapGpuContext *ctx;
    unsigned kind;                 /* 0: encode from pictures, 1: decode texture `index`, 2: encode from textures */
    unsigned n, result;
    /* strided views of the caller's arrays */
    const void **in;
    void **out;
    unsigned long *in_bytes, *out_bytes, *used;
    unsigned *fmts_out, *results;
    /* call-wide arguments */
    unsigned width, height, count, index, flags;
    unsigned long row_bytes;
    const unsigned *formats, *compressors, *chunk_counts;
    const unsigned long *tex_bytes;
} share;

static void *run_share(void *p)
{
    share *s = (share *)p;
    if (s->n == 0) {
        s->result = HapResult_No_Error;
    } else if (s->kind == 0) {
        s->result = HapGpuEncodeFramesRGBA(s->ctx, s->n, s->in, s->width, s->height, s->row_bytes, s->count, s->formats, s->compressors,
                                           s->chunk_counts, s->out, s->out_bytes, s->used, s->results, s->flags);
    } else if (s->kind == 2) {
        s->result = HapGpuEncodeFrames(s->ctx, s->n, s->count, s->in, s->tex_bytes, s->formats, s->compressors, s->chunk_counts, s->out,
                                       s->out_bytes, s->used, s->results, s->flags);
    } else {
        s->result = HapGpuDecodeFrames(s->ctx, s->n, s->in, s->in_bytes, s->index, s->out, s->out_bytes, s->used, s->fmts_out, s->results,
                                       s->flags);
    }
    return NULL;
}

/* frames of context c: c, c + N, c + 2N, ... */
static unsigned share_size(unsigned frames, unsigned c, unsigned n_ctx) { return frames > c ? (frames - c + n_ctx - 1u) / n_ctx : 0u; }

static unsigned run_over_contexts(HapGpuContext *const *contexts, unsigned n_ctx, unsigned kind, unsigned frame_count, unsigned per_frame_in,
                                  const void *const *inputs, const unsigned long *inputs_bytes, void *const *outputs,
                                  const unsigned long *outputs_bytes, unsigned long *used, unsigned *fmts_out, unsigned *results,
                                  const share *common)
{
    share *sh;
    pthread_t *threads;
    unsigned char *started;
    unsigned c, f, first_error = HapResult_No_Error;
    if (!contexts || n_ctx == 0 || !results)
        return HapResult_Bad_Arguments;
    for (c = 0; c < n_ctx; c++)
        if (!contexts[c])
            return HapResult_Bad_Arguments;
    if (frame_count == 0)
        return HapResult_No_Error;
    if (!inputs || !outputs || !outputs_bytes || (kind == 1 && !inputs_bytes) || (kind != 1 && !used)) {
        for (f = 0; f < frame_count; f++)
            results[f] = HapResult_Bad_Arguments;
        return HapResult_Bad_Arguments;
    }
    sh = (share *)calloc(n_ctx, sizeof(*sh));
    threads = (pthread_t *)calloc(n_ctx, sizeof(*threads));
    started = (unsigned char *)calloc(n_ctx, 1);
    if (!sh || !threads || !started) {
        free(sh); free(threads); free(started);
        for (f = 0; f < frame_count; f++)
            results[f] = HapResult_Internal_Error;
        return HapResult_Internal_Error;
    }
    for (c = 0; c < n_ctx; c++) {
        share *s = &sh[c];
        const unsigned n = share_size(frame_count, c, n_ctx);
        unsigned k;
        *s = *common;
        s->ctx = contexts[c];
        s->kind = kind;
        s->n = n;
        if (n == 0)
            continue;
        s->in = (const void **)malloc(sizeof(void *) * (size_t)n * per_frame_in);
        s->out = (void **)malloc(sizeof(void *) * n);
        s->in_bytes = (unsigned long *)calloc(n, sizeof(unsigned long));
        s->out_bytes = (unsigned long *)calloc(n, sizeof(unsigned long));
        s->used = (unsigned long *)calloc(n, sizeof(unsigned long));
        s->fmts_out = (unsigned *)calloc(n, sizeof(unsigned));
        s->results = (unsigned *)calloc(n, sizeof(unsigned));
        if (!s->in || !s->out || !s->in_bytes || !s->out_bytes || !s->used || !s->fmts_out || !s->results) {
            s->result = HapResult_Internal_Error;
            s->n = 0;
            for (k = 0; k < n; k++)
                results[c + (size_t)k * n_ctx] = HapResult_Internal_Error;
            continue;
        }
        for (k = 0; k < n; k++) {
            const size_t fr = c + (size_t)k * n_ctx;
            unsigned t;
            for (t = 0; t < per_frame_in; t++)
                s->in[(size_t)k * per_frame_in + t] = inputs[fr * per_frame_in + t];
            s->out[k] = outputs[fr];
            s->out_bytes[k] = outputs_bytes[fr];
            if (inputs_bytes && kind == 1)
                s->in_bytes[k] = inputs_bytes[fr];
        }
    }
    /* one host thread per context with work (the first share runs on the caller's thread) */
    for (c = 1; c < n_ctx; c++)
        if (sh[c].n && pthread_create(&threads[c], NULL, run_share, &sh[c]) == 0)
            started[c] = 1;
    run_share(&sh[0]);
    for (c = 1; c < n_ctx; c++) {
        if (started[c])
            pthread_join(threads[c], NULL);
        else if (sh[c].n)
            run_share(&sh[c]);                      /* (no thread to be had: the share still gets done) */
    }
    for (c = 0; c < n_ctx; c++) {
        share *s = &sh[c];
        unsigned k;
        for (k = 0; k < s->n; k++) {
            const size_t fr = c + (size_t)k * n_ctx;
            results[fr] = s->results[k];
            if (used)
                used[fr] = s->used[k];
            if (fmts_out)
                fmts_out[fr] = s->fmts_out[k];
        }
        free(s->in); free(s->out); free(s->in_bytes); free(s->out_bytes); free(s->used); free(s->fmts_out); free(s->results);
    }
    /* the call's result: the first failure in FRAME order, as the single-context calls report it */
    for (f = 0; f < frame_count; f++)
        if (results[f] != HapResult_No_Error) {
            first_error = results[f];
            break;
        }
    if (first_error == HapResult_No_Error)
        for (c = 0; c < n_ctx; c++)
            if (sh[c].result != HapResult_No_Error) {
                first_error = sh[c].result;
                break;
            }
    free(sh); free(threads); free(started);
    return first_error;
}

unsigned int HapGpuEncodeFramesRGBAOnDevices(HapGpuContext *const *contexts, unsigned int contextCount, unsigned int frameCount,
                                             const void *const *rgbaFrames, unsigned int width, unsigned int height,
                                             unsigned long rowBytes, unsigned int count, const unsigned int *textureFormats,
                                             const unsigned int *compressors, const unsigned int *chunkCounts,
                                             void *const *outputBuffers, const unsigned long *outputBuffersBytes,
                                             unsigned long *outputBuffersBytesUsed, unsigned int *results, unsigned int flags)
{
    share common;
    memset(&common, 0, sizeof(common));
    common.width = width;
    common.height = height;
    common.row_bytes = rowBytes;
    common.count = count;
    common.formats = textureFormats;
    common.compressors = compressors;
    common.chunk_counts = chunkCounts;
    common.flags = flags;
    return run_over_contexts(contexts, contextCount, 0u, frameCount, 1u, rgbaFrames, NULL, outputBuffers, outputBuffersBytes,
                             outputBuffersBytesUsed, NULL, results, &common);
}

unsigned int HapGpuEncodeFramesOnDevices(HapGpuContext *const *contexts, unsigned int contextCount, unsigned int frameCount,
                                         unsigned int count, const void *const *inputBuffers, const unsigned long *inputBuffersBytes,
                                         const unsigned int *textureFormats, const unsigned int *compressors,
                                         const unsigned int *chunkCounts, void *const *outputBuffers,
                                         const unsigned long *outputBuffersBytes, unsigned long *outputBuffersBytesUsed,
                                         unsigned int *results, unsigned int flags)
{
    share common;
    if (count == 0 || count > 2) {
        unsigned f;
        for (f = 0; results && f < frameCount; f++)
            results[f] = HapResult_Bad_Arguments;
        return HapResult_Bad_Arguments;
    }
    memset(&common, 0, sizeof(common));
    common.count = count;
    common.formats = textureFormats;
    common.compressors = compressors;
    common.chunk_counts = chunkCounts;
    common.tex_bytes = inputBuffersBytes;
    common.flags = flags;
    return run_over_contexts(contexts, contextCount, 2u, frameCount, count, inputBuffers, NULL, outputBuffers, outputBuffersBytes,
                             outputBuffersBytesUsed, NULL, results, &common);
}

unsigned int HapGpuDecodeFramesOnDevices(HapGpuContext *const *contexts, unsigned int contextCount, unsigned int frameCount,
                                         const void *const *inputBuffers, const unsigned long *inputBuffersBytes, unsigned int index,
                                         void *const *outputBuffers, const unsigned long *outputBuffersBytes,
                                         unsigned long *outputBuffersBytesUsed, unsigned int *outputTextureFormats,
                                         unsigned int *results, unsigned int flags)
{
    share common;
    memset(&common, 0, sizeof(common));
    common.index = index;
    common.flags = flags;
    return run_over_contexts(contexts, contextCount, 1u, frameCount, 1u, inputBuffers, inputBuffersBytes, outputBuffers,
                             outputBuffersBytes, outputBuffersBytesUsed, outputTextureFormats, results, &common);
}
