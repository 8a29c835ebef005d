/* percall_loop.c -- bench.py's per-call legs: a plain C client of hap.h that calls HapEncode / HapDecode once per
 * frame, the way a drop-in codec does (reference hap.h:98-104, 132-137), so that the per-call figures carry no
 * Python or ctypes time.  Compiled by bench.py at run time against hap_amd/libhap_amd.so; not part of the product. */
#define _POSIX_C_SOURCE 200809L
#include <time.h>
#include "hap.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* what a single-threaded client passes: hap.h:113-130 */
static void serial_callback(HapDecodeWorkFunction fn, void *p, unsigned int count, void *info)
{
    unsigned int i;
    (void)info;
    for (i = 0; i < count; i++)
        fn(p, i);
}

/* HapDecode of texture `index` of every frame, one call per frame, `reps` times; seconds, or -(first failing result) */
double percall_decode(const void *const *frames, const unsigned long *frame_bytes, unsigned n, unsigned index,
                      void *const *outputs, unsigned long output_bytes, unsigned reps)
{
    unsigned r, f;
    const double t0 = now_s();
    for (r = 0; r < reps; r++)
        for (f = 0; f < n; f++) {
            unsigned long used = 0;
            unsigned int fmt = 0;
            const unsigned rc = HapDecode(frames[f], frame_bytes[f], index, serial_callback, 0, outputs[f], output_bytes, &used, &fmt);
            if (rc != HapResult_No_Error)
                return -(double)rc;
        }
    return now_s() - t0;
}

/* HapEncode of every frame (count textures each: textures[f * count + t]), one call per frame */
double percall_encode(const void *const *textures, const unsigned long *texture_bytes, unsigned count,
                      const unsigned *formats, const unsigned *compressors, const unsigned *chunks, unsigned n,
                      void *const *outputs, unsigned long output_bytes, unsigned long *used, unsigned reps)
{
    unsigned r, f;
    const double t0 = now_s();
    for (r = 0; r < reps; r++)
        for (f = 0; f < n; f++) {
            const unsigned rc = HapEncode(count, (const void **)(textures + (unsigned long)f * count), (unsigned long *)texture_bytes,
                                          (unsigned int *)formats, (unsigned int *)compressors, (unsigned int *)chunks,
                                          outputs[f], output_bytes, &used[f]);
            if (rc != HapResult_No_Error)
                return -(double)rc;
        }
    return now_s() - t0;
}
