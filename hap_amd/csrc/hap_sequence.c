/*
 * hap_sequence.c -- frame-sequence file (include/hap_sequence.h) and the
 * double-buffered disk -> pinned memory -> GPU decode pipeline.
 * Pure C: the GPU work goes through hapb_decode like every other decode call.
 */
#define _FILE_OFFSET_BITS 64
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "../../include/hap_sequence.h"
#include "hap_batch.h"

#define SEQ_HEADER_BYTES 64u
static const char seq_magic[8] = {'H', 'A', 'P', 'S', 'E', 'Q', '1', '\0'};

enum { P_SEQ0 = 8, P_SEQ1 = 9 };   /* pinned scratch slots of the two read-ahead buffers */

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put64(uint8_t *p, uint64_t v) { put32(p, (uint32_t)v); put32(p + 4, (uint32_t)(v >> 32)); }
static uint32_t get32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t get64(const uint8_t *p) { return (uint64_t)get32(p) | ((uint64_t)get32(p + 4) << 32); }

static int write_all(int fd, const void *buf, size_t n)
{
    const uint8_t *p = (const uint8_t *)buf;
    while (n) {
        ssize_t w = write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR)
                continue;
            return -1;
        }
        p += w;
        n -= (size_t)w;
    }
    return 0;
}

static int pread_all(int fd, void *buf, size_t n, uint64_t at)
{
    uint8_t *p = (uint8_t *)buf;
    while (n) {
        ssize_t r = pread(fd, p, n, (off_t)at);
        if (r < 0) {
            if (errno == EINTR)
                continue;
            return -1;
        }
        if (r == 0)
            return -1;          /* file shorter than its index says */
        p += r;
        at += (uint64_t)r;
        n -= (size_t)r;
    }
    return 0;
}

/* ------------------------------------------------------------------ writer -- */
struct HapSequenceWriter {
    int fd;
    uint32_t width, height, rate_num, rate_den;
    uint64_t cursor;
    uint64_t *offsets;      /* count + 1 valid entries */
    uint32_t count, cap;
};

unsigned int HapSequenceWriterOpen(const char *path, unsigned int width, unsigned int height,
                                   unsigned int rateNumerator, unsigned int rateDenominator,
                                   HapSequenceWriter **writer)
{
    HapSequenceWriter *w;
    uint8_t header[SEQ_HEADER_BYTES];
    if (!path || !writer)
        return HapResult_Bad_Arguments;
    *writer = NULL;
    w = (HapSequenceWriter *)calloc(1, sizeof(*w));
    if (!w)
        return HapResult_Internal_Error;
    w->fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (w->fd < 0) {
        free(w);
        return HapResult_Internal_Error;
    }
    w->width = width; w->height = height; w->rate_num = rateNumerator; w->rate_den = rateDenominator;
    w->cap = 256;
    w->offsets = (uint64_t *)malloc(sizeof(uint64_t) * (w->cap + 1u));
    memset(header, 0, sizeof(header));      /* completed by Close */
    if (!w->offsets || write_all(w->fd, header, sizeof(header)) != 0) {
        close(w->fd);
        free(w->offsets);
        free(w);
        return HapResult_Internal_Error;
    }
    w->cursor = SEQ_HEADER_BYTES;
    w->offsets[0] = w->cursor;
    *writer = w;
    return HapResult_No_Error;
}

unsigned int HapSequenceWriterAppend(HapSequenceWriter *w, const void *frame, unsigned long frameBytes)
{
    if (!w || !frame || frameBytes == 0 || w->fd < 0)
        return HapResult_Bad_Arguments;
    if (w->count == 0xFFFFFFFFu)
        return HapResult_Bad_Arguments;
    if (w->count == w->cap) {
        uint32_t cap = w->cap * 2u;
        uint64_t *n = (uint64_t *)realloc(w->offsets, sizeof(uint64_t) * ((size_t)cap + 1u));
        if (!n)
            return HapResult_Internal_Error;
        w->offsets = n;
        w->cap = cap;
    }
    if (write_all(w->fd, frame, frameBytes) != 0)
        return HapResult_Internal_Error;
    w->cursor += frameBytes;
    w->count += 1u;
    w->offsets[w->count] = w->cursor;
    return HapResult_No_Error;
}

unsigned int HapSequenceWriterClose(HapSequenceWriter *w)
{
    unsigned result = HapResult_No_Error;
    uint8_t header[SEQ_HEADER_BYTES];
    uint8_t *index;
    uint32_t i;
    if (!w)
        return HapResult_Bad_Arguments;
    index = (uint8_t *)malloc(8u * ((size_t)w->count + 1u));
    if (!index) {
        result = HapResult_Internal_Error;
    } else {
        for (i = 0; i <= w->count; i++)
            put64(index + 8u * i, w->offsets[i]);
        memset(header, 0, sizeof(header));
        memcpy(header, seq_magic, 8);
        put32(header + 8, 1u);
        put32(header + 12, w->width);
        put32(header + 16, w->height);
        put32(header + 20, w->rate_num);
        put32(header + 24, w->rate_den);
        put32(header + 28, w->count);
        put64(header + 32, w->cursor);
        if (write_all(w->fd, index, 8u * ((size_t)w->count + 1u)) != 0 ||
            pwrite(w->fd, header, sizeof(header), 0) != (ssize_t)sizeof(header))
            result = HapResult_Internal_Error;
        free(index);
    }
    if (close(w->fd) != 0)
        result = HapResult_Internal_Error;
    free(w->offsets);
    free(w);
    return result;
}

/* ------------------------------------------------------------------ reader -- */
struct HapSequenceReader {
    int fd;
    uint32_t width, height, rate_num, rate_den, count;
    uint64_t *offsets;      /* count + 1 */
};

unsigned int HapSequenceReaderOpen(const char *path, HapSequenceReader **reader)
{
    HapSequenceReader *r;
    uint8_t header[SEQ_HEADER_BYTES];
    uint8_t *index = NULL;
    uint64_t index_at, file_bytes;
    struct stat st;
    uint32_t i;
    unsigned result = HapResult_Bad_Frame;
    if (!path || !reader)
        return HapResult_Bad_Arguments;
    *reader = NULL;
    r = (HapSequenceReader *)calloc(1, sizeof(*r));
    if (!r)
        return HapResult_Internal_Error;
    r->fd = open(path, O_RDONLY);
    if (r->fd < 0 || fstat(r->fd, &st) != 0) {
        result = HapResult_Internal_Error;
        goto fail;
    }
    file_bytes = (uint64_t)st.st_size;
    if (file_bytes < SEQ_HEADER_BYTES || pread_all(r->fd, header, sizeof(header), 0) != 0)
        goto fail;
    if (memcmp(header, seq_magic, 8) != 0 || get32(header + 8) != 1u)
        goto fail;
    r->width = get32(header + 12);
    r->height = get32(header + 16);
    r->rate_num = get32(header + 20);
    r->rate_den = get32(header + 24);
    r->count = get32(header + 28);
    index_at = get64(header + 32);
    if (index_at < SEQ_HEADER_BYTES || index_at > file_bytes ||
        (file_bytes - index_at) / 8u < (uint64_t)r->count + 1u)
        goto fail;
    index = (uint8_t *)malloc(8u * ((size_t)r->count + 1u));
    r->offsets = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)r->count + 1u));
    if (!index || !r->offsets) {
        result = HapResult_Internal_Error;
        goto fail;
    }
    if (pread_all(r->fd, index, 8u * ((size_t)r->count + 1u), index_at) != 0)
        goto fail;
    for (i = 0; i <= r->count; i++) {
        r->offsets[i] = get64(index + 8u * i);
        /* frames lie between the header and the index, in order, none empty or beyond 4 GiB */
        if (r->offsets[i] < SEQ_HEADER_BYTES || r->offsets[i] > index_at ||
            (i && (r->offsets[i] <= r->offsets[i - 1] || r->offsets[i] - r->offsets[i - 1] > 0xFFFFFFFFull)))
            goto fail;
    }
    free(index);
    *reader = r;
    return HapResult_No_Error;
fail:
    free(index);
    if (r->fd >= 0)
        close(r->fd);
    free(r->offsets);
    free(r);
    return result;
}

void HapSequenceReaderClose(HapSequenceReader *r)
{
    if (!r)
        return;
    close(r->fd);
    free(r->offsets);
    free(r);
}

unsigned int HapSequenceReaderInfo(const HapSequenceReader *r, unsigned int *width, unsigned int *height,
                                   unsigned int *rateNumerator, unsigned int *rateDenominator,
                                   unsigned int *frameCount)
{
    if (!r)
        return HapResult_Bad_Arguments;
    if (width) *width = r->width;
    if (height) *height = r->height;
    if (rateNumerator) *rateNumerator = r->rate_num;
    if (rateDenominator) *rateDenominator = r->rate_den;
    if (frameCount) *frameCount = r->count;
    return HapResult_No_Error;
}

unsigned long HapSequenceReaderFrameBytes(const HapSequenceReader *r, unsigned int frame)
{
    if (!r || frame >= r->count)
        return 0;
    return (unsigned long)(r->offsets[frame + 1u] - r->offsets[frame]);
}

unsigned int HapSequenceReaderRead(HapSequenceReader *r, unsigned int first, unsigned int count,
                                   void *buffer, unsigned long bufferBytes, unsigned long *offsets)
{
    uint64_t begin, total;
    unsigned i;
    if (!r || !buffer || count == 0 || first >= r->count || count > r->count - first)
        return HapResult_Bad_Arguments;
    begin = r->offsets[first];
    total = r->offsets[first + count] - begin;
    if (total > bufferBytes)
        return HapResult_Buffer_Too_Small;
    if (pread_all(r->fd, buffer, (size_t)total, begin) != 0)
        return HapResult_Internal_Error;
    if (offsets)
        for (i = 0; i <= count; i++)
            offsets[i] = (unsigned long)(r->offsets[first + i] - begin);
    return HapResult_No_Error;
}

/* ---------------------------------------------------------------- pipeline -- */
typedef struct read_job {
    HapSequenceReader *reader;
    unsigned first, count;
    void *buffer;
    unsigned long buffer_bytes;
    unsigned result;
} read_job;

static void *read_main(void *arg)
{
    read_job *j = (read_job *)arg;
    j->result = HapSequenceReaderRead(j->reader, j->first, j->count, j->buffer, j->buffer_bytes, NULL);
    return NULL;
}

unsigned int HapGpuDecodeSequence(HapGpuContext *ctx, HapSequenceReader *r, unsigned int first, unsigned int count,
                                  unsigned int index, unsigned int batch, void *const *outputs,
                                  const unsigned long *output_bytes, unsigned long *output_used,
                                  unsigned int *output_formats, unsigned int *results)
{
    unsigned first_error = HapResult_No_Error, done, b, batches;
    unsigned long biggest = 0;
    void *pinned[2] = {NULL, NULL};
    const void **ptrs = NULL;
    unsigned long *lens = NULL;
    read_job job;
    pthread_t thread;
    int thread_live = 0;

    if (!ctx || !r || !outputs || !output_bytes || !results || count == 0 || first >= r->count ||
        count > r->count - first || index > 1)
        return HapResult_Bad_Arguments;
    if (batch == 0)
        batch = 16;
    if (batch > count)
        batch = count;
    if (batch > 32768u)
        batch = 32768u;                                 /* (one launch sequence: grid dimensions hold at most 65535) */
    batches = (count + batch - 1u) / batch;
    for (b = 0; b < batches; b++) {
        const unsigned f0 = first + b * batch, n = (count - b * batch) < batch ? (count - b * batch) : batch;
        const uint64_t bytes = r->offsets[f0 + n] - r->offsets[f0];
        if (bytes > 0x7FFFFFFFFFFFull)
            return HapResult_Bad_Arguments;
        if (bytes > biggest)
            biggest = (unsigned long)bytes;
    }
    ptrs = (const void **)malloc(sizeof(void *) * batch);
    lens = (unsigned long *)malloc(sizeof(unsigned long) * batch);
    if (!ptrs || !lens) {
        free(ptrs); free(lens);
        return HapResult_Internal_Error;
    }
    /* the context stays locked for the whole call: the two read-ahead buffers are its scratch and must not be
       resized by another thread's call while the helper thread fills them */
    hapgpu_rt_lock(ctx->rt);
    pinned[0] = hapgpu_rt_pinned_scratch(ctx->rt, P_SEQ0, biggest);
    pinned[1] = batches > 1 ? hapgpu_rt_pinned_scratch(ctx->rt, P_SEQ1, biggest) : pinned[0];
    if (!pinned[0] || !pinned[1]) {
        hapgpu_rt_unlock(ctx->rt);
        free(ptrs); free(lens);
        return HapResult_Internal_Error;
    }

    /* first batch: nothing to overlap with */
    job.reader = r; job.first = first; job.count = batch; job.buffer = pinned[0]; job.buffer_bytes = biggest;
    read_main(&job);
    done = 0;
    for (b = 0; b < batches; b++) {
        const unsigned f0 = first + b * batch, n = (count - b * batch) < batch ? (count - b * batch) : batch;
        const uint8_t *base = (const uint8_t *)pinned[b & 1u];
        unsigned i, rc;
        if (thread_live) {
            pthread_join(thread, NULL);
            thread_live = 0;
        }
        if (job.result != HapResult_No_Error) {       /* the read of THIS batch failed */
            for (i = 0; i < count - done; i++)
                results[done + i] = HapResult_Internal_Error;
            first_error = first_error ? first_error : job.result;
            break;
        }
        for (i = 0; i < n; i++) {
            ptrs[i] = base + (r->offsets[f0 + i] - r->offsets[f0]);
            lens[i] = (unsigned long)(r->offsets[f0 + i + 1u] - r->offsets[f0 + i]);
        }
        /* read-ahead of the next batch into the other buffer while the GPU works on this one */
        if (b + 1u < batches) {
            const unsigned nf = f0 + n, nn = (count - (b + 1u) * batch) < batch ? (count - (b + 1u) * batch) : batch;
            job.reader = r; job.first = nf; job.count = nn; job.buffer = pinned[(b + 1u) & 1u]; job.buffer_bytes = biggest;
            job.result = HapResult_Internal_Error;
            if (pthread_create(&thread, NULL, read_main, &job) == 0)
                thread_live = 1;
            else
                read_main(&job);
        }
        rc = hapb_decode(ctx, n, ptrs, lens, index, outputs + done, output_bytes + done,
                         output_used ? output_used + done : NULL, output_formats ? output_formats + done : NULL,
                         results + done, 0, NULL, NULL);
        if (rc != HapResult_No_Error && first_error == HapResult_No_Error)
            first_error = rc;
        done += n;
    }
    if (thread_live)
        pthread_join(thread, NULL);
    hapgpu_rt_unlock(ctx->rt);
    free(ptrs);
    free(lens);
    return first_error;
}

/* ------------------------------------------------------- encode pipeline -- */
typedef struct write_job {
    HapSequenceWriter *writer;
    unsigned count;
    const uint8_t *base;
    size_t stride;
    const unsigned long *used;
    unsigned result;
    unsigned written;           /* frames of the batch that were appended */
} write_job;

static void *write_main(void *arg)
{
    write_job *j = (write_job *)arg;
    unsigned i;
    j->result = HapResult_No_Error;
    j->written = 0;
    for (i = 0; i < j->count && j->result == HapResult_No_Error; i++) {
        j->result = HapSequenceWriterAppend(j->writer, j->base + j->stride * i, j->used[i]);
        if (j->result == HapResult_No_Error)
            j->written = i + 1u;
    }
    return NULL;
}

/* frames [first, first + count) were encoded; the first `written` of them are in the file: only those report their
   size, the others a failure (their own, or Internal_Error for a frame that was fine but never written) */
static void publish_batch(unsigned int *results, unsigned long *frame_bytes, unsigned first, unsigned count, unsigned written,
                          const unsigned *res, const unsigned long *used)
{
    unsigned i;
    for (i = 0; i < count; i++) {
        const int in_file = i < written && res[i] == HapResult_No_Error;
        if (results)
            results[first + i] = in_file ? (unsigned)HapResult_No_Error
                                         : res[i] != HapResult_No_Error ? res[i] : (unsigned)HapResult_Internal_Error;
        if (frame_bytes)
            frame_bytes[first + i] = in_file ? used[i] : 0ul;
    }
}

unsigned int HapGpuEncodeSequence(HapGpuContext *ctx, HapSequenceWriter *w, unsigned int count,
                                  const void *const *rgba_frames, unsigned int width, unsigned int height,
                                  unsigned long row_bytes, unsigned int texture_count, const unsigned int *formats,
                                  const unsigned int *compressors, const unsigned int *chunk_counts,
                                  unsigned int flags, unsigned int batch, unsigned long *frame_bytes,
                                  unsigned int *results)
{
    unsigned first_error = HapResult_No_Error, done = 0, b, batches, i;
    unsigned long lengths[2] = {0, 0}, cap;
    size_t stride;
    uint8_t *pinned[2] = {NULL, NULL};
    void **outs = NULL;
    unsigned long *caps = NULL, *used[2] = {NULL, NULL};
    unsigned *res[2] = {NULL, NULL};
    write_job job;
    pthread_t thread;
    int thread_live = 0;
    unsigned pending_first = 0, pending_slot = 0;       /* the batch the helper thread is appending */

    if (!ctx || !w || !rgba_frames || count == 0 || texture_count == 0 || texture_count > 2 || !formats || !compressors ||
        !chunk_counts || width == 0 || height == 0 || (width & 3u) || (height & 3u))
        return HapResult_Bad_Arguments;
    for (i = 0; i < texture_count; i++) {
        const unsigned long block = (formats[i] == HapTextureFormat_RGB_DXT1 || formats[i] == HapTextureFormat_A_RGTC1) ? 8ul : 16ul;
        lengths[i] = (unsigned long)(width / 4u) * (height / 4u) * block;
    }
    cap = HapMaxEncodedLength(texture_count, lengths, (unsigned int *)formats, (unsigned int *)chunk_counts);
    if (cap == 0)
        return HapResult_Bad_Arguments;
    if (batch == 0)
        batch = 16;
    if (batch > count)
        batch = count;
    if (batch > 32768u)
        batch = 32768u;                                 /* (one launch sequence: grid dimensions hold at most 65535) */
    batches = (count + batch - 1u) / batch;
    stride = ((size_t)cap + 255u) & ~(size_t)255u;
    outs = (void **)malloc(sizeof(void *) * batch);
    caps = (unsigned long *)malloc(sizeof(unsigned long) * batch);
    used[0] = (unsigned long *)calloc(batch, sizeof(unsigned long));
    used[1] = (unsigned long *)calloc(batch, sizeof(unsigned long));
    res[0] = (unsigned *)malloc(sizeof(unsigned) * batch);
    res[1] = (unsigned *)malloc(sizeof(unsigned) * batch);
    if (!outs || !caps || !used[0] || !used[1] || !res[0] || !res[1]) {
        free(outs); free(caps); free(used[0]); free(used[1]); free(res[0]); free(res[1]);
        return HapResult_Internal_Error;
    }
    /* (the context stays locked for the whole call: the two frame buffers are its scratch, see HapGpuDecodeSequence) */
    hapgpu_rt_lock(ctx->rt);
    pinned[0] = (uint8_t *)hapgpu_rt_pinned_scratch(ctx->rt, P_SEQ0, stride * batch);
    pinned[1] = batches > 1 ? (uint8_t *)hapgpu_rt_pinned_scratch(ctx->rt, P_SEQ1, stride * batch) : pinned[0];
    if (!pinned[0] || !pinned[1]) {
        hapgpu_rt_unlock(ctx->rt);
        free(outs); free(caps); free(used[0]); free(used[1]); free(res[0]); free(res[1]);
        return HapResult_Internal_Error;
    }
    memset(&job, 0, sizeof(job));
    for (i = 0; results && i < count; i++)
        results[i] = HapResult_Internal_Error;          /* (what a frame keeps when the call ends before its batch) */
    for (i = 0; frame_bytes && i < count; i++)
        frame_bytes[i] = 0ul;
    for (b = 0; b < batches && first_error == HapResult_No_Error; b++) {
        const unsigned n = (count - b * batch) < batch ? (count - b * batch) : batch;
        uint8_t *base = pinned[b & 1u];
        unsigned rc;
        for (i = 0; i < n; i++) {
            outs[i] = base + stride * i;
            caps[i] = cap;
        }
        /* (the helper may still be writing batch b - 1 from the OTHER buffer: the GPU fills this one meanwhile) */
        rc = hapb_encode_rgba(ctx, n, rgba_frames + done, width, height, row_bytes, texture_count, formats, compressors,
                              chunk_counts, outs, caps, used[b & 1u], res[b & 1u], flags);
        if (thread_live) {
            pthread_join(thread, NULL);
            thread_live = 0;
            publish_batch(results, frame_bytes, pending_first, job.count, job.written, res[pending_slot], used[pending_slot]);
            if (job.result != HapResult_No_Error)
                first_error = job.result;
        }
        for (i = 0; i < n; i++)
            if (res[b & 1u][i] != HapResult_No_Error && rc == HapResult_No_Error)
                rc = res[b & 1u][i];
        if (rc != HapResult_No_Error && first_error == HapResult_No_Error)
            first_error = rc;
        if (first_error != HapResult_No_Error) {
            /* nothing of this batch goes to the file: a frame that failed, or the batch before it could not be written */
            publish_batch(results, frame_bytes, done, n, 0u, res[b & 1u], used[b & 1u]);
            break;
        }
        job.writer = w; job.count = n; job.base = base; job.stride = stride; job.used = used[b & 1u];
        job.result = HapResult_Internal_Error;
        job.written = 0;
        pending_first = done;
        pending_slot = b & 1u;
        if (b + 1u < batches && pthread_create(&thread, NULL, write_main, &job) == 0) {
            thread_live = 1;
        } else {
            write_main(&job);
            publish_batch(results, frame_bytes, done, n, job.written, res[b & 1u], used[b & 1u]);
            if (job.result != HapResult_No_Error)
                first_error = job.result;
        }
        done += n;
    }
    if (thread_live) {
        pthread_join(thread, NULL);
        publish_batch(results, frame_bytes, pending_first, job.count, job.written, res[pending_slot], used[pending_slot]);
        if (job.result != HapResult_No_Error && first_error == HapResult_No_Error)
            first_error = job.result;
    }
    hapgpu_rt_unlock(ctx->rt);
    free(outs); free(caps); free(used[0]); free(used[1]); free(res[0]); free(res[1]);
    return first_error;
}
