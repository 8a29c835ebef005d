/*
 * hap_api.c -- the exported symbols: the six functions of include/hap.h (same
 * names and signatures as the reference, /root/reference/source/hap.h:76-152)
 * and the context / batch functions of include/hap_gpu.h.  Pure C99.
 */
#include "hap_batch.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ contexts -- */

unsigned int HapGpuCreate(int device, HapGpuContext **context)
{
    HapGpuContext *c;
    int rc;
    if (!context)
        return HapResult_Bad_Arguments;
    *context = NULL;
    c = (HapGpuContext *)calloc(1, sizeof(*c));
    if (!c)
        return HapResult_Internal_Error;
    rc = hapgpu_rt_create(device, &c->rt);
    if (rc) {
        free(c);
        return rc == 1 ? HapResult_Bad_Arguments : HapResult_Internal_Error;
    }
    c->frag_log2 = 13;   /* 8 KiB: best decode occupancy for ~1 % more bytes than 16 KiB */
    {
        const char *e = getenv("HAP_AMD_FRAGMENT_LOG2");
        if (e && atoi(e) >= 10 && atoi(e) <= 16)
            c->frag_log2 = (unsigned)atoi(e);
    }
    {
        const char *e = getenv("HAP_AMD_BYTE_GRANULAR");
        c->byte_granular = (e && atoi(e) != 0) ? 1u : 0u;
        c->position_lanes = HAP_AB_ENV("HAP_AMD_POSITION_LANES") ? 1u : 0u;
        c->no_half_tiles = HAP_AB_ENV("HAP_AMD_NO_HALF_TILES") ? 1u : 0u;
        c->no_block_scan = getenv("HAP_AMD_NO_BLOCK_SCAN") ? 1u : 0u;
        c->no_fusion = getenv("HAP_AMD_NO_FUSION") ? 1u : 0u;
        c->no_placing = getenv("HAP_AMD_NO_PLACING") ? 1u : 0u;
        c->placing_holdoff_calls = getenv("HAP_AMD_PLACING_HOLDOFF") ? (unsigned)atoi(getenv("HAP_AMD_PLACING_HOLDOFF")) : 8u;
        c->placing_min_frames = getenv("HAP_AMD_PLACING_MIN_FRAMES") ? (unsigned)atoi(getenv("HAP_AMD_PLACING_MIN_FRAMES")) : 8u;
        /* RGTC1 planes of large textures go through the [2, 6] field kernel (block-per-lane decodable: 2.9x the decode
           rate at the same size); HAP_AMD_RGTC1_LAYOUT overrides: 0 = position-per-lane compressor, 44 = [4, 4] */
        c->rgtc1_fields = 26u;
        if (HAP_AB_ENV("HAP_AMD_RGTC1_LAYOUT")) {
            const char *layout = HAP_AB_ENV("HAP_AMD_RGTC1_LAYOUT");
            const int v = layout ? atoi(layout) : 26;
            if (v == 0 || v == 26 || v == 44)
                c->rgtc1_fields = (unsigned)v;
        }
    }
    *context = c;
    return HapResult_No_Error;
}

void HapGpuDestroy(HapGpuContext *context)
{
    if (!context)
        return;
    /* an encode call begun and never finished: its launches are waited for (hapgpu_rt_destroy synchronises), its results
       are NOT written -- the client's arrays may be gone by now */
    if (context->pending_encode) {
        hapb_encode_abandon(context, context->pending_encode);
        context->pending_encode = NULL;
    }
    hapgpu_rt_destroy(context->rt);
    free(context);
}

static pthread_once_t g_default_once = PTHREAD_ONCE_INIT;
static HapGpuContext *g_default;

static void make_default(void)
{
    const char *e = getenv("HAP_AMD_DEVICE");
    if (HapGpuCreate(e ? atoi(e) : -1, &g_default) != HapResult_No_Error) {
        g_default = NULL;
        fprintf(stderr, "hap_amd: cannot create a GPU context; HapEncode/HapDecode will fail "
                        "(this library has no CPU path)\n");
    }
}

HapGpuContext *HapGpuDefaultContext(void)
{
    pthread_once(&g_default_once, make_default);
    return g_default;
}

/* hap.h has no context argument, and the reference is re-entrant (no globals in hap.c): a client may call
 * HapDecode from several threads at once, or again from inside its HapDecodeCallback (hap.h:113-130) while the
 * outer call is still waiting for it.  A context serves one call at a time, so the hap.h entry points take
 * whichever default context is free and add one (same device, own stream and scratch) when all are busy. */
#define HAP_DEFAULT_POOL 8
#define HAP_BATCH_SLICE 32768u      /* frames per launch sequence: grid dimensions y / z hold at most 65535 */
static HapGpuContext *g_pool[HAP_DEFAULT_POOL];
static pthread_t g_pool_owner[HAP_DEFAULT_POOL];     /* the thread inside a hap.h call on that member, if g_pool_busy */
static unsigned char g_pool_busy[HAP_DEFAULT_POOL];
static unsigned g_pool_count;
static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;

/* returns a default context with its lock HELD, or NULL when there is no GPU -- or when every member of the pool is
   in use by calls of THIS thread (HapDecode nested HAP_DEFAULT_POOL deep through callbacks): waiting would never end */
static HapGpuContext *acquire_default_context(void)
{
    HapGpuContext *first = HapGpuDefaultContext(), *c = NULL;
    const pthread_t self = pthread_self();
    unsigned i;
    int wait_for = -1;
    if (!first)
        return NULL;
    pthread_mutex_lock(&g_pool_lock);
    if (g_pool_count == 0)
        g_pool[g_pool_count++] = first;
    for (i = 0; i < g_pool_count && !c; i++)
        if (hapgpu_rt_trylock(g_pool[i]->rt) == 0) {
            c = g_pool[i];
            g_pool_owner[i] = self;
            g_pool_busy[i] = 1;
        }
    if (c)
        c->frag_log2 = first->frag_log2;       /* (HapGpuSetFragmentLog2 on the default context reaches every member) */
    if (!c && g_pool_count < HAP_DEFAULT_POOL &&
        HapGpuCreate(hapgpu_rt_device(first->rt), &c) == HapResult_No_Error) {
        c->frag_log2 = first->frag_log2;
        g_pool_owner[g_pool_count] = self;
        g_pool_busy[g_pool_count] = 1;
        g_pool[g_pool_count++] = c;
        hapgpu_rt_lock(c->rt);
    }
    if (!c) {
        /* pool exhausted: wait for a member that some other thread (or a client holding the default context's handle)
           is using */
        for (i = 0; i < g_pool_count; i++) {
            if (!(g_pool_busy[i] && pthread_equal(g_pool_owner[i], self)) && wait_for < 0)
                wait_for = (int)i;
        }
    }
    pthread_mutex_unlock(&g_pool_lock);
    if (!c && wait_for >= 0) {
        c = g_pool[wait_for];
        hapgpu_rt_lock(c->rt);
        pthread_mutex_lock(&g_pool_lock);
        g_pool_owner[wait_for] = self;
        g_pool_busy[wait_for] = 1;
        c->frag_log2 = first->frag_log2;
        pthread_mutex_unlock(&g_pool_lock);
    }
    return c;
}

static void release_default_context(HapGpuContext *c)
{
    unsigned i;
    pthread_mutex_lock(&g_pool_lock);
    for (i = 0; i < g_pool_count; i++)
        if (g_pool[i] == c)
            g_pool_busy[i] = 0;
    pthread_mutex_unlock(&g_pool_lock);
    hapgpu_rt_unlock(c->rt);
}

unsigned int HapGpuSetFragmentLog2(HapGpuContext *context, unsigned int log2_bytes)
{
    if (!context || log2_bytes < 10 || log2_bytes > 16)
        return HapResult_Bad_Arguments;
    context->frag_log2 = log2_bytes;
    return HapResult_No_Error;
}

unsigned long HapGpuTableFallbackCount(HapGpuContext *context)
{
    unsigned long n;
    if (!context)
        return 0;
    hapgpu_rt_lock(context->rt);
    n = context->table_fallbacks;
    hapgpu_rt_unlock(context->rt);
    if (context == g_default) {                /* the hap.h entry points may have run on any member of the pool */
        unsigned i;
        pthread_mutex_lock(&g_pool_lock);
        for (i = 0; i < g_pool_count; i++)
            if (g_pool[i] != context)
                n += g_pool[i]->table_fallbacks;
        pthread_mutex_unlock(&g_pool_lock);
    }
    return n;
}

unsigned long HapGpuResolvedBlockCount(HapGpuContext *context)
{
    unsigned long n;
    if (!context)
        return 0;
    hapgpu_rt_lock(context->rt);
    n = hapgpu_rt_resolved_blocks(context->rt);
    hapgpu_rt_unlock(context->rt);
    return n;
}

unsigned long HapGpuPlacementRetryCount(HapGpuContext *context)
{
    unsigned long n;
    if (!context)
        return 0;
    hapgpu_rt_lock(context->rt);
    n = context->placement_retries;
    hapgpu_rt_unlock(context->rt);
    return n;
}

unsigned int HapGpuFineChunkCount(unsigned long textureBytes, unsigned int textureFormat)
{
    unsigned long want;
    if (textureBytes == 0 || textureBytes > 0xFFFFFFFFul || hapf_nibble_from_format(textureFormat) == 0)
        return 0;
    want = (textureBytes + 8191ul) / 8192ul;
    if (textureBytes < 16ul)
        return 1;
    {
        /* The SMALLEST divisor of the block count that is at least bytes / 8 KiB: chunks of at most one fragment (ADVICE r05:
           walking DOWN to a divisor gave 1080p DXT5 240 chunks of 8640 bytes -- two fragments each, which the table-less
           road into the block-per-lane decoder does not take).  Where the block count has no divisor up to four times that
           many chunks (a prime number of blocks), the largest one below it, as for any chunk count (hap.c:277-300). */
        const unsigned long block = (textureFormat == HapTextureFormat_RGB_DXT1 || textureFormat == HapTextureFormat_A_RGTC1) ? 8ul : 16ul;
        const unsigned long blocks = textureBytes / block;
        unsigned long c, top = want * 4ul;
        if (top > 3355431ul)
            top = 3355431ul;
        if (top > blocks)
            top = blocks;
        for (c = want; c <= top; c++)
            if (c != 0ul && blocks % c == 0ul)
                return (unsigned)c;
    }
    return hapf_limit_chunk_count(textureBytes, textureFormat, want > 3355431ul ? 3355431u : (unsigned)want);
}

unsigned long HapGpuPlacementTimeoutCount(HapGpuContext *context)
{
    unsigned long n;
    if (!context)
        return 0;
    hapgpu_rt_lock(context->rt);
    n = context->placement_timeouts;
    hapgpu_rt_unlock(context->rt);
    return n;
}

unsigned int HapGpuSynchronize(HapGpuContext *context)
{
    unsigned r;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    r = hapgpu_rt_sync(context->rt) ? HapResult_Internal_Error : HapResult_No_Error;
    hapgpu_rt_unlock(context->rt);
    return r;
}

/* --------------------------------------------------------------- hap.h -- */

/* reference hap.c:324-353 */
unsigned long HapMaxEncodedLength(unsigned int count, unsigned long *lengths,
                                  unsigned int *textureFormats, unsigned int *chunkCounts)
{
    unsigned long total = 8;
    unsigned i;
    if (count == 0 || count > 2 || !lengths || !textureFormats || !chunkCounts)
        return 0;
    for (i = 0; i < count; i++) {
        if (chunkCounts[i] == 0)
            return 0;
        total += (unsigned long)hapf_texture_bound(lengths[i], textureFormats[i], HapCompressorSnappy, chunkCounts[i]);
    }
    return total;
}

static unsigned env_encode_flags(void)
{
    const char *e = getenv("HAP_AMD_FRAGMENT_INDEX"), *c = getenv("HAP_AMD_COARSE_MATCHES"), *s = getenv("HAP_AMD_SMALLER_FILES");
    /* Frames from plain hap.h HapEncode carry nothing the Hap specification does not name: the private fragment table
       (section 0x46 inside the Decode Instructions Container) is written on request only, HAP_AMD_FRAGMENT_INDEX=1 --
       the reference skips unknown sections there (hap.c:701-703), but a drop-in library cannot know that every other
       parser of its frames does.  The batched HapGpu* calls take it as a flag. */
    return ((e && atoi(e) != 0) ? HAPGPU_ENCODE_FRAGMENT_INDEX : 0u) |
           ((c && atoi(c) != 0) ? HAPGPU_ENCODE_COARSE_MATCHES : 0u) |
           ((s && atoi(s) != 0) ? HAPGPU_ENCODE_SMALLER_FILES : 0u);
}

/* reference hap.c:506-604 */
unsigned int HapEncode(unsigned int count, const void **inputBuffers, unsigned long *inputBuffersBytes,
                       unsigned int *textureFormats, unsigned int *compressors, unsigned int *chunkCounts,
                       void *outputBuffer, unsigned long outputBufferBytes,
                       unsigned long *outputBufferBytesUsed)
{
    HapGpuContext *ctx;
    unsigned result = HapResult_Internal_Error, rc, flags, i;
    unsigned long used = 0;
    void *out = outputBuffer;
    if (count == 0 || count > 2 || !inputBuffers || !inputBuffersBytes || !textureFormats || !compressors ||
        !chunkCounts || !outputBuffer || outputBufferBytes == 0 || !outputBufferBytesUsed)
        return HapResult_Bad_Arguments;
    flags = env_encode_flags();
    /* A texture whose bytes do not divide by its chunk count is no real block texture; the reference's chunked form
       then drops the remainder (hap.c:433), and whether a frame takes that form or the raw one must not depend on
       the private table's few bytes: such textures are written without it, exactly as the reference decides. */
    for (i = 0; i < count; i++) {
        const unsigned n = chunkCounts[i] ? hapf_limit_chunk_count(inputBuffersBytes[i], textureFormats[i], chunkCounts[i]) : 0u;
        if (n == 0u || inputBuffersBytes[i] % n != 0u)
            flags &= ~HAPGPU_ENCODE_FRAGMENT_INDEX;
    }
    ctx = acquire_default_context();
    if (!ctx)
        return HapResult_Internal_Error;
    rc = hapb_encode(ctx, 1, count, (const void *const *)inputBuffers, inputBuffersBytes, textureFormats,
                     compressors, chunkCounts, &out, &outputBufferBytes, &used, &result, flags, 0);
    release_default_context(ctx);
    if (rc == HapResult_No_Error)
        *outputBufferBytesUsed = used;
    return rc;
}

/* reference hap.c:993-1040 */
unsigned int HapDecode(const void *inputBuffer, unsigned long inputBufferBytes, unsigned int index,
                       HapDecodeCallback callback, void *info, void *outputBuffer,
                       unsigned long outputBufferBytes, unsigned long *outputBufferBytesUsed,
                       unsigned int *outputBufferTextureFormat)
{
    HapGpuContext *ctx;
    unsigned result = HapResult_Internal_Error, rc, fmt = 0;
    unsigned long used = 0;
    if (!inputBuffer || index > 1 || !callback || !outputBuffer || !outputBufferTextureFormat)
        return HapResult_Bad_Arguments;
    ctx = acquire_default_context();
    if (!ctx)
        return HapResult_Internal_Error;
    rc = hapb_decode(ctx, 1, &inputBuffer, &inputBufferBytes, index, &outputBuffer, &outputBufferBytes, &used,
                     &fmt, &result, 0, callback, info);
    release_default_context(ctx);
    /* the reference stores the format as soon as the section type has been read (hap.c:754) */
    *outputBufferTextureFormat = fmt;
    if (rc == HapResult_No_Error && outputBufferBytesUsed)
        *outputBufferBytesUsed = used;
    return rc;
}

/* The reference object also exports two helpers that hap.h does not declare (they are non-static in
 * hap.c:732 and hap.c:932); kept for link compatibility with clients that reach for them. */
int hap_get_section_at_index(const void *input_buffer, uint32_t input_buffer_bytes, unsigned int index,
                             const void **section, uint32_t *section_length, unsigned int *section_type)
{
    hapf_reader r;
    uint64_t off = 0;
    unsigned rc;
    hapf_reader_init_host(&r, input_buffer, input_buffer_bytes);
    *section = NULL;
    *section_length = 0;
    *section_type = 0;
    rc = hapf_locate(&r, input_buffer_bytes, index, &off, section_length, section_type);
    if (rc == HapResult_No_Error)
        *section = (const uint8_t *)input_buffer + off;
    hapf_reader_free(&r);
    return (int)rc;
}

unsigned int hap_decode_single_texture(const void *texture_section, uint32_t texture_section_length,
                                       unsigned int texture_section_type, HapDecodeCallback callback, void *info,
                                       void *outputBuffer, unsigned long outputBufferBytes,
                                       unsigned long *outputBufferBytesUsed, unsigned int *outputBufferTextureFormat)
{
    /* re-wrap the bare section in an 8-byte top-level header and take the normal path (host memory only) */
    uint8_t *frame;
    unsigned rc;
    if (!texture_section || !callback || !outputBuffer || !outputBufferTextureFormat)
        return HapResult_Bad_Arguments;
    frame = (uint8_t *)malloc((size_t)texture_section_length + 8u);
    if (!frame)
        return HapResult_Internal_Error;
    hapf_write_section(frame, 8u, texture_section_length, texture_section_type);
    memcpy(frame + 8, texture_section, texture_section_length);
    rc = HapDecode(frame, (unsigned long)texture_section_length + 8ul, 0, callback, info, outputBuffer,
                   outputBufferBytes, outputBufferBytesUsed, outputBufferTextureFormat);
    free(frame);
    return rc;
}

/* Inspectors read section headers only.  Host frames are parsed in place; for
 * device-resident frames the few header bytes are copied back. */
typedef struct inspect_fetch {
    HapGpuContext *ctx;
    const uint8_t *frame;
} inspect_fetch;

static int inspect_fetch_cb(void *user, uint64_t offset, uint64_t length, uint8_t *dst)
{
    inspect_fetch *f = (inspect_fetch *)user;
    int rc;
    hapgpu_rt_lock(f->ctx->rt);
    rc = hapgpu_rt_d2h(f->ctx->rt, dst, f->frame + offset, (size_t)length);
    if (!rc)
        rc = hapgpu_rt_sync(f->ctx->rt);
    hapgpu_rt_unlock(f->ctx->rt);
    return rc;
}

static void open_reader(hapf_reader *r, inspect_fetch *f, const void *frame, unsigned long bytes)
{
    /* classify without forcing a context into existence for plain host memory users */
    HapGpuContext *ctx = g_default;
    if (ctx && hapgpu_rt_is_device_ptr(ctx->rt, frame)) {
        hapf_reader_init_host(r, NULL, 0);
        f->ctx = ctx;
        f->frame = (const uint8_t *)frame;
        r->fetch = inspect_fetch_cb;
        r->user = f;
    } else {
        hapf_reader_init_host(r, frame, bytes);
    }
}

/* reference hap.c:1042-1087 (no NULL checks there either) */
unsigned int HapGetFrameTextureCount(const void *inputBuffer, unsigned long inputBufferBytes,
                                     unsigned int *outputTextureCount)
{
    hapf_reader r;
    inspect_fetch f;
    unsigned rc;
    open_reader(&r, &f, inputBuffer, inputBufferBytes);
    rc = hapf_texture_count(&r, inputBufferBytes, outputTextureCount);
    hapf_reader_free(&r);
    return rc;
}

/* reference hap.c:1089-1126 */
unsigned int HapGetFrameTextureFormat(const void *inputBuffer, unsigned long inputBufferBytes,
                                      unsigned int index, unsigned int *outputBufferTextureFormat)
{
    hapf_reader r;
    inspect_fetch f;
    uint64_t off;
    uint32_t len;
    unsigned type = 0, rc;
    if (!inputBuffer || index > 1 || !outputBufferTextureFormat)
        return HapResult_Bad_Arguments;
    open_reader(&r, &f, inputBuffer, inputBufferBytes);
    rc = hapf_locate(&r, (uint32_t)inputBufferBytes, index, &off, &len, &type);
    hapf_reader_free(&r);
    if (rc != HapResult_No_Error)
        return rc;
    *outputBufferTextureFormat = hapf_format_from_nibble(type & 0xFu);
    return *outputBufferTextureFormat ? HapResult_No_Error : HapResult_Bad_Frame;
}

/* reference hap.c:1128-1188 */
unsigned int HapGetFrameTextureChunkCount(const void *inputBuffer, unsigned long inputBufferBytes,
                                          unsigned int index, int *chunk_count)
{
    hapf_reader r;
    inspect_fetch f;
    hapf_texture_plan plan;
    *chunk_count = 0;                     /* before validation, like hap.c:1134 */
    if (!inputBuffer || index > 1)
        return HapResult_Bad_Arguments;
    open_reader(&r, &f, inputBuffer, inputBufferBytes);
    hapf_plan_texture(&r, (uint32_t)inputBufferBytes, index, 0, &plan);
    *chunk_count = plan.chunk_count;
    hapf_plan_free(&plan);
    hapf_reader_free(&r);
    return plan.result;
}

/* ----------------------------------------------------------- hap_gpu.h -- */

unsigned int HapGpuCompressRGBA(HapGpuContext *context, const void *rgba, unsigned int width,
                                unsigned int height, unsigned long rowBytes, unsigned int textureFormat,
                                void *output, unsigned long outputBytes, unsigned long *outputBytesUsed)
{
    unsigned r;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    r = hapb_compress_rgba(context, rgba, width, height, rowBytes, textureFormat, output, outputBytes,
                           outputBytesUsed, 1);
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuDecompressRGBA(HapGpuContext *context, const void *texture, unsigned long textureBytes,
                                  unsigned int textureFormat, const void *alphaTexture, unsigned long alphaBytes,
                                  unsigned int width, unsigned int height, void *rgba, unsigned long rowBytes)
{
    unsigned r;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    r = hapb_decompress_rgba(context, texture, textureBytes, textureFormat, alphaTexture, alphaBytes, width, height,
                             rgba, rowBytes);
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuEncodeFrames(HapGpuContext *context, unsigned int frameCount, unsigned int count,
                                const void *const *inputBuffers, const unsigned long *inputBuffersBytes,
                                const unsigned int *textureFormats, const unsigned int *compressors,
                                const unsigned int *chunkCounts, void *const *outputBuffers,
                                const unsigned long *outputBuffersBytes, unsigned long *outputBuffersBytesUsed,
                                unsigned int *results, unsigned int flags)
{
    unsigned r = HapResult_No_Error, done;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    /* (the kernels index frames with a grid dimension that holds 65535: larger batches go in slices) */
    if (frameCount <= HAP_BATCH_SLICE || !inputBuffers || !outputBuffers || !outputBuffersBytes || !outputBuffersBytesUsed || !results ||
        count == 0 || count > 2) {
        r = hapb_encode(context, frameCount, count, inputBuffers, inputBuffersBytes, textureFormats, compressors,
                        chunkCounts, outputBuffers, outputBuffersBytes, outputBuffersBytesUsed, results, flags, 0);
    } else {
        for (done = 0; done < frameCount; done += HAP_BATCH_SLICE) {
            const unsigned n = frameCount - done < HAP_BATCH_SLICE ? frameCount - done : HAP_BATCH_SLICE;
            const unsigned rc = hapb_encode(context, n, count, inputBuffers + (size_t)done * count, inputBuffersBytes, textureFormats,
                                            compressors, chunkCounts, outputBuffers + done, outputBuffersBytes + done,
                                            outputBuffersBytesUsed + done, results + done, flags, 0);
            if (r == HapResult_No_Error)
                r = rc;
        }
    }
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuEncodeFramesRGBA(HapGpuContext *context, unsigned int frameCount,
                                    const void *const *rgbaFrames, unsigned int width, unsigned int height,
                                    unsigned long rowBytes, unsigned int count,
                                    const unsigned int *textureFormats, const unsigned int *compressors,
                                    const unsigned int *chunkCounts, void *const *outputBuffers,
                                    const unsigned long *outputBuffersBytes,
                                    unsigned long *outputBuffersBytesUsed, unsigned int *results,
                                    unsigned int flags)
{
    unsigned r = HapResult_No_Error, done;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    if (frameCount <= HAP_BATCH_SLICE || !rgbaFrames || !outputBuffers || !outputBuffersBytes || !outputBuffersBytesUsed || !results) {
        r = hapb_encode_rgba(context, frameCount, rgbaFrames, width, height, rowBytes, count, textureFormats,
                             compressors, chunkCounts, outputBuffers, outputBuffersBytes, outputBuffersBytesUsed,
                             results, flags);
    } else {
        for (done = 0; done < frameCount; done += HAP_BATCH_SLICE) {
            const unsigned n = frameCount - done < HAP_BATCH_SLICE ? frameCount - done : HAP_BATCH_SLICE;
            const unsigned rc = hapb_encode_rgba(context, n, rgbaFrames + done, width, height, rowBytes, count, textureFormats,
                                                 compressors, chunkCounts, outputBuffers + done, outputBuffersBytes + done,
                                                 outputBuffersBytesUsed + done, results + done, flags);
            if (r == HapResult_No_Error)
                r = rc;
        }
    }
    hapgpu_rt_unlock(context->rt);
    return r;
}

/* The two halves of HapGpuEncodeFramesRGBA / HapGpuEncodeFrames (include/hap_gpu.h): everything launched, nothing waited for. */
unsigned int HapGpuEncodeFramesRGBABegin(HapGpuContext *context, unsigned int frameCount,
                                         const void *const *rgbaFrames, unsigned int width, unsigned int height,
                                         unsigned long rowBytes, unsigned int count,
                                         const unsigned int *textureFormats, const unsigned int *compressors,
                                         const unsigned int *chunkCounts, void *const *outputBuffers,
                                         const unsigned long *outputBuffersBytes,
                                         unsigned long *outputBuffersBytesUsed, unsigned int *results,
                                         unsigned int flags)
{
    unsigned r;
    if (!context || frameCount > HAP_BATCH_SLICE)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    context->defer_encode = 1u;
    r = hapb_encode_rgba(context, frameCount, rgbaFrames, width, height, rowBytes, count, textureFormats,
                         compressors, chunkCounts, outputBuffers, outputBuffersBytes, outputBuffersBytesUsed,
                         results, flags);
    context->defer_encode = 0u;
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuEncodeFramesBegin(HapGpuContext *context, unsigned int frameCount, unsigned int count,
                                     const void *const *inputBuffers, const unsigned long *inputBuffersBytes,
                                     const unsigned int *textureFormats, const unsigned int *compressors,
                                     const unsigned int *chunkCounts, void *const *outputBuffers,
                                     const unsigned long *outputBuffersBytes,
                                     unsigned long *outputBuffersBytesUsed, unsigned int *results,
                                     unsigned int flags)
{
    unsigned r;
    if (!context || frameCount > HAP_BATCH_SLICE)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    context->defer_encode = 1u;
    r = hapb_encode(context, frameCount, count, inputBuffers, inputBuffersBytes, textureFormats, compressors, chunkCounts,
                    outputBuffers, outputBuffersBytes, outputBuffersBytesUsed, results, flags, 0);
    context->defer_encode = 0u;
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuEncodeFramesFinish(HapGpuContext *context)
{
    unsigned r = HapResult_No_Error;
    HapbEncodePending *pd;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    pd = context->pending_encode;
    context->pending_encode = NULL;
    if (pd)
        r = hapb_encode_complete(context, pd);
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuDecodeFrames(HapGpuContext *context, unsigned int frameCount,
                                const void *const *inputBuffers, const unsigned long *inputBuffersBytes,
                                unsigned int index, void *const *outputBuffers,
                                const unsigned long *outputBuffersBytes, unsigned long *outputBuffersBytesUsed,
                                unsigned int *outputTextureFormats, unsigned int *results, unsigned int flags)
{
    unsigned r = HapResult_No_Error, done;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    if (frameCount <= HAP_BATCH_SLICE || !inputBuffers || !inputBuffersBytes || !outputBuffers || !outputBuffersBytes || !results) {
        r = hapb_decode(context, frameCount, inputBuffers, inputBuffersBytes, index, outputBuffers,
                        outputBuffersBytes, outputBuffersBytesUsed, outputTextureFormats, results, flags, NULL, NULL);
    } else {
        for (done = 0; done < frameCount; done += HAP_BATCH_SLICE) {
            const unsigned n = frameCount - done < HAP_BATCH_SLICE ? frameCount - done : HAP_BATCH_SLICE;
            const unsigned rc = hapb_decode(context, n, inputBuffers + done, inputBuffersBytes + done, index, outputBuffers + done,
                                            outputBuffersBytes + done, outputBuffersBytesUsed ? outputBuffersBytesUsed + done : NULL,
                                            outputTextureFormats ? outputTextureFormats + done : NULL, results + done, flags, NULL, NULL);
            if (r == HapResult_No_Error)
                r = rc;
        }
    }
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuDecodeFrameTextures(HapGpuContext *context, unsigned int frameCount,
                                       const void *const *inputBuffers, const unsigned long *inputBuffersBytes,
                                       unsigned int textureCount, void *const *outputBuffers,
                                       const unsigned long *outputBuffersBytes, unsigned long *outputBuffersBytesUsed,
                                       unsigned int *outputTextureFormats, unsigned int *results, unsigned int flags)
{
    unsigned r = HapResult_No_Error, f, t;
    const void **inputs;
    unsigned long *bytes;
    unsigned *indices;
    size_t entries, done;
    if (!context || !results || textureCount == 0 || textureCount > 2)
        return HapResult_Bad_Arguments;
    if (frameCount == 0)
        return HapResult_No_Error;
    entries = (size_t)frameCount * textureCount;
    if (!inputBuffers || !inputBuffersBytes || !outputBuffers || !outputBuffersBytes) {
        for (done = 0; done < entries; done++)
            results[done] = HapResult_Bad_Arguments;
        return HapResult_Bad_Arguments;
    }
    /* entry f * textureCount + t = texture t of frame f: one batch, so the frames' headers come to the host once
       and every texture's units go out in the same launches */
    inputs = (const void **)malloc(sizeof(*inputs) * entries);
    bytes = (unsigned long *)malloc(sizeof(*bytes) * entries);
    indices = (unsigned *)malloc(sizeof(*indices) * entries);
    if (!inputs || !bytes || !indices) {
        free(inputs); free(bytes); free(indices);
        for (done = 0; done < entries; done++)
            results[done] = HapResult_Internal_Error;
        return HapResult_Internal_Error;
    }
    for (f = 0; f < frameCount; f++)
        for (t = 0; t < textureCount; t++) {
            inputs[(size_t)f * textureCount + t] = inputBuffers[f];
            bytes[(size_t)f * textureCount + t] = inputBuffersBytes[f];
            indices[(size_t)f * textureCount + t] = t;
        }
    hapgpu_rt_lock(context->rt);
    for (done = 0; done < entries; done += HAP_BATCH_SLICE) {
        const unsigned n = (unsigned)(entries - done < HAP_BATCH_SLICE ? entries - done : HAP_BATCH_SLICE);
        unsigned rc;
        context->decode_indices = indices + done;
        rc = hapb_decode(context, n, inputs + done, bytes + done, 0, outputBuffers + done, outputBuffersBytes + done,
                         outputBuffersBytesUsed ? outputBuffersBytesUsed + done : NULL,
                         outputTextureFormats ? outputTextureFormats + done : NULL, results + done, flags, NULL, NULL);
        if (r == HapResult_No_Error)
            r = rc;
    }
    hapgpu_rt_unlock(context->rt);
    free(inputs); free(bytes); free(indices);
    return r;
}

unsigned int HapGpuDecodeFramesRGBA(HapGpuContext *context, unsigned int frameCount,
                                    const void *const *inputBuffers, const unsigned long *inputBuffersBytes,
                                    unsigned int textureCount, void *const *rgbaFrames, unsigned int width,
                                    unsigned int height, unsigned long rowBytes, unsigned int *results,
                                    unsigned int flags)
{
    unsigned r;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    r = hapb_decode_rgba(context, frameCount, inputBuffers, inputBuffersBytes, textureCount, rgbaFrames, width, height,
                         rowBytes, results, flags);
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuJoinChunkGroupsDevice(HapGpuContext *context, unsigned int groupCount,
                                         const void *const *groupFrames, const unsigned long *groupFramesBytes,
                                         void *outputBuffer, unsigned long outputBufferBytes,
                                         unsigned long *outputBufferBytesUsed)
{
    unsigned r;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    r = hapb_join_device(context, groupCount, groupFrames, groupFramesBytes, outputBuffer, outputBufferBytes,
                         outputBufferBytesUsed);
    hapgpu_rt_unlock(context->rt);
    return r;
}

/* callback that "runs" one contiguous group of chunks (hap.h:113-130 lets a client run any subset) */
typedef struct chunk_group {
    unsigned first, count;
} chunk_group;

static void run_chunk_group(HapDecodeWorkFunction function, void *p, unsigned int count, void *info)
{
    const chunk_group *g = (const chunk_group *)info;
    unsigned i;
    for (i = g->first; i < count && i - g->first < g->count; i++)
        function(p, i);
}

unsigned int HapGpuDecodeChunkGroup(HapGpuContext *context, const void *inputBuffer, unsigned long inputBufferBytes,
                                    unsigned int index, unsigned int firstChunk, unsigned int chunkCount,
                                    void *outputBuffer, unsigned long outputBufferBytes,
                                    unsigned long *outputBufferBytesUsed, unsigned int *outputBufferTextureFormat)
{
    chunk_group g;
    unsigned result = HapResult_Internal_Error, rc, fmt = 0;
    unsigned long used = 0;
    if (!context || !inputBuffer || index > 1 || !outputBuffer)
        return HapResult_Bad_Arguments;
    g.first = firstChunk;
    g.count = chunkCount;
    hapgpu_rt_lock(context->rt);
    rc = hapb_decode(context, 1, &inputBuffer, &inputBufferBytes, index, &outputBuffer, &outputBufferBytes, &used,
                     &fmt, &result, 0, run_chunk_group, &g);
    hapgpu_rt_unlock(context->rt);
    if (outputBufferTextureFormat)
        *outputBufferTextureFormat = fmt;
    if (rc == HapResult_No_Error && outputBufferBytesUsed)
        *outputBufferBytesUsed = used;
    return rc;
}

/* decoded position of every chunk: the running sum the reference builds in hap.c:794-838 */
unsigned int HapGpuGetFrameTextureChunkLayout(const void *inputBuffer, unsigned long inputBufferBytes,
                                              unsigned int index, unsigned int capacity,
                                              unsigned long *decodedOffsets, unsigned int *chunkCount)
{
    hapf_reader r;
    inspect_fetch f;
    hapf_texture_plan plan;
    unsigned result, n, i;
    unsigned long cursor = 0;
    if (!inputBuffer || index > 1 || !decodedOffsets || !chunkCount)
        return HapResult_Bad_Arguments;
    *chunkCount = 0;
    open_reader(&r, &f, inputBuffer, inputBufferBytes);
    hapf_plan_texture(&r, (uint32_t)inputBufferBytes, index, 1, &plan);
    result = plan.result;
    n = plan.mode == HAPGPU_JOB_COMPLEX ? (unsigned)plan.chunk_count : 1u;
    if (result == HapResult_No_Error) {
        *chunkCount = n;
        if (capacity < n + 1u)
            result = HapResult_Buffer_Too_Small;
    }
    for (i = 0; result == HapResult_No_Error && i < n; i++) {
        uint64_t at = plan.mode == HAPGPU_JOB_COMPLEX ? plan.payload_offset + plan.chunks[i].src_off : plan.section_offset;
        uint32_t len = plan.mode == HAPGPU_JOB_COMPLEX ? plan.chunks[i].src_len : plan.section_length;
        unsigned codec = plan.mode == HAPGPU_JOB_COMPLEX ? (plan.chunks[i].codec & 0xFFu)
                         : plan.mode == HAPGPU_JOB_RAW ? HAP_NIBBLE_NONE : HAP_NIBBLE_SNAPPY;
        decodedOffsets[i] = cursor;
        if (codec == HAP_NIBBLE_NONE) {
            cursor += len;
        } else if (codec == HAP_NIBBLE_SNAPPY) {
            /* the stream's leading varint (snappy_uncompressed_length, hap.c:813) */
            const unsigned take = len < 5u ? len : 5u;
            const uint8_t *v = take ? hapf_need(&r, at, take) : NULL;
            uint64_t value = 0;
            unsigned k = 0, done = 0;
            while (v && k < take && !done) {
                value |= (uint64_t)(v[k] & 0x7Fu) << (7u * k);
                done = !(v[k] & 0x80u);
                k++;
            }
            if (!done || value > 0xFFFFFFFFull)
                result = v || !take ? HapResult_Bad_Frame : HapResult_Internal_Error;
            cursor += (unsigned long)value;
        } else {
            result = HapResult_Bad_Frame;             /* hap.c:637-640 */
        }
    }
    if (result == HapResult_No_Error)
        decodedOffsets[n] = cursor;
    hapf_plan_free(&plan);
    hapf_reader_free(&r);
    return result;
}

unsigned int HapGpuSetProfiling(HapGpuContext *context, unsigned int enable)
{
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    hapgpu_rt_set_profiling(context->rt, (int)enable);
    hapgpu_rt_unlock(context->rt);
    return HapResult_No_Error;
}

unsigned int HapGpuCollectProfileN(HapGpuContext *context, unsigned int classCount, unsigned long *launches, double *milliseconds)
{
    unsigned r;
    if (!context || !launches || !milliseconds)
        return HapResult_Bad_Arguments;
    if (classCount > HapGpuKernel_ClassCount)
        classCount = HapGpuKernel_ClassCount;
    hapgpu_rt_lock(context->rt);
    /* (events of classes beyond the caller's arrays are drained and dropped) */
    r = hapgpu_rt_collect_profile(context->rt, launches, milliseconds, classCount) ? HapResult_Internal_Error : HapResult_No_Error;
    hapgpu_rt_unlock(context->rt);
    return r;
}

/* The entry point of the first header (eight classes, no count argument): a client built against it has arrays of
   eight.  New code says how many entries its arrays have (HapGpuCollectProfileN). */
unsigned int HapGpuCollectProfile(HapGpuContext *context, unsigned long *launches, double *milliseconds)
{
    /* (the class count of the header this entry point was last released with: nine, encode_fused included -- ADVICE r05) */
    return HapGpuCollectProfileN(context, 9u, launches, milliseconds);
}

unsigned int HapGpuTimerStart(HapGpuContext *context)
{
    unsigned r;
    if (!context)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    r = hapgpu_rt_timer_start(context->rt) ? HapResult_Internal_Error : HapResult_No_Error;
    hapgpu_rt_unlock(context->rt);
    return r;
}

unsigned int HapGpuTimerStop(HapGpuContext *context, double *milliseconds)
{
    unsigned r;
    if (!context || !milliseconds)
        return HapResult_Bad_Arguments;
    hapgpu_rt_lock(context->rt);
    r = hapgpu_rt_timer_stop(context->rt, milliseconds) ? HapResult_Internal_Error : HapResult_No_Error;
    hapgpu_rt_unlock(context->rt);
    return r;
}
