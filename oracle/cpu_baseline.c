/*
 * cpu_baseline.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Timing harness for the CPU leg that bench.py reports beside the GPU number
 * ("cpu_baseline").  Compiled twice:
 *   - into liboracle.so           (kind "port": the restatement in this dir)
 *   - into _ref/libhap_ref.so     (kind "reference": -DBASELINE_REFERENCE, the
 *                                  unmodified /root/reference/source/hap.c +
 *                                  libsnappy 1.1.8)
 * Decode fans chunk work out to threads through the HapDecodeCallback contract
 * (hap.h:113-130); encode is serial per frame in the reference (hap.c:448-476)
 * so frames are spread over threads instead.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef BASELINE_REFERENCE
#include "hap.h"
#define API_DECODE HapDecode
#define API_ENCODE HapEncode
#define SYM(name) refbase_##name
typedef HapDecodeWorkFunction work_fn;
#else
#include "oracle.h"
#define API_DECODE ohap_decode
#define API_ENCODE(c, in, nb, f, cp, cc, o, ob, u) \
    ohap_encode(c, in, (const unsigned long *)(nb), f, cp, cc, o, ob, u)
#define SYM(name) oraclebase_##name
typedef OHapWork work_fn;
#endif

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- thread pinning ("threads pinned"; unpinned runs were erratic on the survey box) -------------
 * Worker t of a call runs on the t-th CPU of the process's affinity mask at the time of the call; the calling thread
 * gets its own mask back when the call returns. */
static int g_pin = 1;
static cpu_set_t g_allowed;
static int g_allowed_n;

void SYM(set_pinning)(int on) { g_pin = on; }
int SYM(pinning)(void) { return g_pin; }

static void pin_begin(cpu_set_t *saved)
{
    CPU_ZERO(saved);
    if (!g_pin || pthread_getaffinity_np(pthread_self(), sizeof(*saved), saved) != 0) {
        g_allowed_n = 0;
        return;
    }
    g_allowed = *saved;
    g_allowed_n = CPU_COUNT(&g_allowed);
}

static void pin_self(unsigned tid)
{
    int want, cpu;
    cpu_set_t one;
    if (!g_pin || g_allowed_n <= 0)
        return;
    want = (int)(tid % (unsigned)g_allowed_n);
    for (cpu = 0; cpu < CPU_SETSIZE; cpu++)
        if (CPU_ISSET(cpu, &g_allowed) && want-- == 0)
            break;
    if (cpu >= CPU_SETSIZE)
        return;
    CPU_ZERO(&one);
    CPU_SET(cpu, &one);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
}

static void pin_end(const cpu_set_t *saved)
{
    if (g_pin && g_allowed_n > 0)
        (void)pthread_setaffinity_np(pthread_self(), sizeof(*saved), saved);
}

/* ---- decode: chunk fan-out --------------------------------------------- */
typedef struct { work_fn fn; void *p; unsigned count, tid, threads; } fan_t;

static void *fan_main(void *arg)
{
    fan_t *f = (fan_t *)arg;
    unsigned i;
    for (i = f->tid; i < f->count; i += f->threads)
        f->fn(f->p, i);
    return NULL;
}

static void fan_callback(work_fn fn, void *p, unsigned count, void *info)
{
    unsigned threads = *(unsigned *)info, t;
    pthread_t th[256];
    fan_t job[256];
    if (threads > 256)
        threads = 256;
    if (threads <= 1) {
        for (t = 0; t < count; t++)
            fn(p, t);
        return;
    }
    for (t = 0; t < threads; t++) {
        job[t].fn = fn; job[t].p = p; job[t].count = count; job[t].tid = t; job[t].threads = threads;
        if (t + 1 < threads)
            pthread_create(&th[t], NULL, fan_main, &job[t]);
    }
    fan_main(&job[threads - 1]);
    for (t = 0; t + 1 < threads; t++)
        pthread_join(th[t], NULL);
}

/* Decode texture `index` of every frame, `reps` times; returns seconds, or a
 * negative HapResult on failure.  out must hold the largest decoded texture. */
double SYM(decode)(const void *const *frames, const unsigned long *frame_bytes, unsigned nframes,
                   unsigned index, void *out, unsigned long out_bytes, unsigned threads, unsigned reps)
{
    double t0;
    unsigned r, f;
    t0 = now_s();
    for (r = 0; r < reps; r++)
        for (f = 0; f < nframes; f++) {
            unsigned long used = 0;
            unsigned fmt = 0;
            unsigned rc = API_DECODE(frames[f], frame_bytes[f], index, fan_callback, &threads, out,
                                     out_bytes, &used, &fmt);
            if (rc != 0)
                return -(double)rc;
        }
    return now_s() - t0;
}

/* ---- decode: whole frames per thread (serial callback inside) -------------- */
typedef struct {
    const void *const *frames; const unsigned long *frame_bytes; unsigned nframes, index, tid, threads, reps, rc;
    unsigned char *out; unsigned long out_stride;
} dec_t;

static void serial_callback(work_fn fn, void *p, unsigned count, void *info)
{
    unsigned i;
    (void)info;
    for (i = 0; i < count; i++)
        fn(p, i);
}

static void *dec_main(void *arg)
{
    dec_t *d = (dec_t *)arg;
    unsigned r, f;
    pin_self(d->tid);
    for (r = 0; r < d->reps; r++)
        for (f = d->tid; f < d->nframes; f += d->threads) {
            unsigned long used = 0;
            unsigned fmt = 0;
            unsigned rc = API_DECODE(d->frames[f], d->frame_bytes[f], d->index, serial_callback, NULL,
                                     d->out + (size_t)d->tid * d->out_stride, d->out_stride, &used, &fmt);
            if (rc != 0)
                d->rc = rc;
        }
    return NULL;
}

/* out must hold threads * out_stride bytes */
double SYM(decode_parallel)(const void *const *frames, const unsigned long *frame_bytes, unsigned nframes,
                            unsigned index, void *out, unsigned long out_stride, unsigned threads, unsigned reps)
{
    pthread_t th[256];
    dec_t job[256];
    unsigned t;
    double t0;
    if (threads == 0) threads = 1;
    if (threads > 256) threads = 256;
    cpu_set_t saved;
    pin_begin(&saved);
    t0 = now_s();
    for (t = 0; t < threads; t++) {
        dec_t *d = &job[t];
        d->frames = frames; d->frame_bytes = frame_bytes; d->nframes = nframes; d->index = index;
        d->tid = t; d->threads = threads; d->reps = reps; d->rc = 0;
        d->out = (unsigned char *)out; d->out_stride = out_stride;
        if (t + 1 < threads)
            pthread_create(&th[t], NULL, dec_main, d);
    }
    dec_main(&job[threads - 1]);
    for (t = 0; t + 1 < threads; t++)
        pthread_join(th[t], NULL);
    pin_end(&saved);
    for (t = 0; t < threads; t++)
        if (job[t].rc)
            return -(double)job[t].rc;
    return now_s() - t0;
}

/* ---- encode: one frame per thread ---------------------------------------- */
typedef struct {
    unsigned count, nframes, tid, threads, reps, rc;
    const void *const *inputs;        /* nframes * count texture pointers */
    const unsigned long *input_bytes; /* count */
    const unsigned *formats, *compressors, *chunks;
    unsigned char *out;               /* threads * out_stride */
    unsigned long out_stride;
    unsigned long *used;              /* nframes */
} enc_t;

static void *enc_main(void *arg)
{
    enc_t *e = (enc_t *)arg;
    unsigned r, f;
    pin_self(e->tid);
    for (r = 0; r < e->reps; r++)
        for (f = e->tid; f < e->nframes; f += e->threads) {
            unsigned rc = API_ENCODE(e->count, (const void **)(e->inputs + (size_t)f * e->count),
                                     (unsigned long *)e->input_bytes, (unsigned *)e->formats,
                                     (unsigned *)e->compressors, (unsigned *)e->chunks,
                                     e->out + (size_t)e->tid * e->out_stride, e->out_stride,
                                     &e->used[f]);
            if (rc != 0)
                e->rc = rc;
        }
    return NULL;
}

double SYM(encode)(unsigned count, const void *const *inputs, const unsigned long *input_bytes,
                   const unsigned *formats, const unsigned *compressors, const unsigned *chunks,
                   unsigned nframes, void *out, unsigned long out_stride, unsigned long *used,
                   unsigned threads, unsigned reps)
{
    pthread_t th[256];
    enc_t job[256];
    unsigned t;
    double t0;
    if (threads == 0)
        threads = 1;
    if (threads > 256)
        threads = 256;
    cpu_set_t saved;
    pin_begin(&saved);
    t0 = now_s();
    for (t = 0; t < threads; t++) {
        enc_t *e = &job[t];
        e->count = count; e->nframes = nframes; e->tid = t; e->threads = threads; e->reps = reps;
        e->rc = 0; e->inputs = inputs; e->input_bytes = input_bytes; e->formats = formats;
        e->compressors = compressors; e->chunks = chunks; e->out = (unsigned char *)out;
        e->out_stride = out_stride; e->used = used;
        if (t + 1 < threads)
            pthread_create(&th[t], NULL, enc_main, e);
    }
    enc_main(&job[threads - 1]);
    for (t = 0; t + 1 < threads; t++)
        pthread_join(th[t], NULL);
    pin_end(&saved);
    for (t = 0; t < threads; t++)
        if (job[t].rc)
            return -(double)job[t].rc;
    return now_s() - t0;
}

#ifndef BASELINE_REFERENCE
/* ---- block encode (ours only: the reference has no RGBA->DXT stage) ------- */
typedef struct {
    const unsigned char *rgba; unsigned w, h0, h1; size_t row_bytes; unsigned char *out;
    unsigned format, reps, tid;
} bc_t;

static void *bc_main(void *arg)
{
    bc_t *b = (bc_t *)arg;
    unsigned rows = b->h1 - b->h0, r;
    size_t bpb = (b->format == 0x83F0 || b->format == 0x8DBB) ? 8 : 16;
    const unsigned char *src = b->rgba + (size_t)b->h0 * b->row_bytes;
    unsigned char *dst = b->out + (size_t)(b->h0 / 4) * (b->w / 4) * bpb;
    pin_self(b->tid);
    for (r = 0; r < b->reps; r++) {
        if (b->format == 0x83F0) obc_encode_dxt1(src, b->w, rows, b->row_bytes, dst);
        else if (b->format == 0x83F3) obc_encode_dxt5(src, b->w, rows, b->row_bytes, dst);
        else if (b->format == 0x01) obc_encode_ycocg_dxt5(src, b->w, rows, b->row_bytes, dst);
        else obc_encode_rgtc1_alpha(src, b->w, rows, b->row_bytes, dst);
    }
    return NULL;
}

double oraclebase_bc_encode(const void *rgba, unsigned w, unsigned h, size_t row_bytes,
                            unsigned format, void *out, unsigned threads, unsigned reps)
{
    pthread_t th[256];
    bc_t job[256];
    unsigned t, block_rows = h / 4;
    double t0;
    if (threads == 0) threads = 1;
    if (threads > 256) threads = 256;
    if (threads > block_rows) threads = block_rows;
    {
    cpu_set_t saved;
    pin_begin(&saved);
    t0 = now_s();
    for (t = 0; t < threads; t++) {
        job[t].tid = t;
        job[t].rgba = (const unsigned char *)rgba; job[t].w = w; job[t].row_bytes = row_bytes;
        job[t].out = (unsigned char *)out; job[t].format = format; job[t].reps = reps;
        job[t].h0 = 4 * (unsigned)((unsigned long)block_rows * t / threads);
        job[t].h1 = 4 * (unsigned)((unsigned long)block_rows * (t + 1) / threads);
        if (t + 1 < threads)
            pthread_create(&th[t], NULL, bc_main, &job[t]);
    }
    bc_main(&job[threads - 1]);
    for (t = 0; t + 1 < threads; t++)
        pthread_join(th[t], NULL);
    pin_end(&saved);
    }
    return now_s() - t0;
}
#endif
