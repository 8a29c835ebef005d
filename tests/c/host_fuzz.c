/* Sanitizer harness for the host-side container code (hap_frame.c, hap_join.c, the file part of hap_sequence.c):
 * built with -fsanitize=address,undefined by tests/test_host_sanitizers.py and fed frames made by the checker;
 * it mutates them a few thousand times and runs every parser over the result.  Any out-of-bounds access,
 * leak or undefined operation fails the test; return values are irrelevant here. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hap_batch.h"
#include "../../include/hap_sequence.h"

/* the GPU side is not linked: the few symbols hap_sequence.c needs are stubs that must never be reached */
unsigned hapb_decode(HapGpuContext *ctx, unsigned frame_count, const void *const *inputs,
                     const unsigned long *input_bytes, unsigned index, void *const *outputs,
                     const unsigned long *output_bytes, unsigned long *output_used, unsigned *output_formats,
                     unsigned *results, unsigned flags, HapDecodeCallback callback, void *callback_info)
{
    (void)ctx; (void)frame_count; (void)inputs; (void)input_bytes; (void)index; (void)outputs; (void)output_bytes;
    (void)output_used; (void)output_formats; (void)results; (void)flags; (void)callback; (void)callback_info;
    abort();
}
unsigned hapb_encode_rgba(HapGpuContext *ctx, unsigned frame_count, const void *const *rgba_frames,
                          unsigned width, unsigned height, unsigned long row_bytes, unsigned count,
                          const unsigned *formats, const unsigned *compressors, const unsigned *chunk_counts,
                          void *const *outputs, const unsigned long *output_bytes,
                          unsigned long *output_used, unsigned *results, unsigned flags)
{
    (void)ctx; (void)frame_count; (void)rgba_frames; (void)width; (void)height; (void)row_bytes; (void)count;
    (void)formats; (void)compressors; (void)chunk_counts; (void)outputs; (void)output_bytes; (void)output_used;
    (void)results; (void)flags;
    abort();
}
unsigned long HapMaxEncodedLength(unsigned int count, unsigned long *lengths, unsigned int *formats, unsigned int *chunks)
{
    (void)count; (void)lengths; (void)formats; (void)chunks;
    abort();
}
void hapgpu_rt_lock(hapgpu_rt *rt) { (void)rt; abort(); }
void hapgpu_rt_unlock(hapgpu_rt *rt) { (void)rt; abort(); }
void *hapgpu_rt_pinned_scratch(hapgpu_rt *rt, int slot, size_t bytes) { (void)rt; (void)slot; (void)bytes; abort(); }

static uint64_t rng_state = 0x48415031u;
static uint32_t rnd(void)
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

static void exercise(const uint8_t *frame, size_t n, const uint8_t *other, size_t other_n)
{
    unsigned index;
    unsigned count = 0;
    hapf_reader r;
    /* exact-size heap copy: the sanitizer sees every byte read past the end */
    uint8_t *copy = (uint8_t *)malloc(n ? n : 1);
    memcpy(copy, frame, n);
    hapf_reader_init_host(&r, copy, n);
    (void)hapf_texture_count(&r, (unsigned long)n, &count);
    hapf_reader_free(&r);
    for (index = 0; index < 3; index++) {
        int want;
        for (want = 0; want < 2; want++) {
            hapf_texture_plan plan;
            hapf_reader_init_host(&r, copy, n);
            hapf_plan_texture(&r, (uint32_t)n, index, want, &plan);
            hapf_plan_free(&plan);
            hapf_reader_free(&r);
        }
    }
    {
        const void *groups[2];
        unsigned long sizes[2], used = 0;
        uint8_t *out = (uint8_t *)malloc(n + other_n + 64);
        groups[0] = copy; sizes[0] = (unsigned long)n;
        groups[1] = other; sizes[1] = (unsigned long)other_n;
        (void)HapGpuJoinChunkGroups(2, groups, sizes, out, (unsigned long)(n + other_n + 64), &used);
        (void)HapGpuJoinChunkGroups(1, groups, sizes, out, (unsigned long)n, &used);
        free(out);
    }
    free(copy);
}

int main(int argc, char **argv)
{
    int a, rounds = 1500;
    const char *seq_path = NULL;
    for (a = 1; a < argc; a++) {
        FILE *f;
        long n;
        uint8_t *base, *work;
        int it;
        if (a == 1) {               /* first argument: scratch path for the sequence-file checks */
            seq_path = argv[a];
            continue;
        }
        f = fopen(argv[a], "rb");
        if (!f)
            return 2;
        fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
        base = (uint8_t *)malloc((size_t)n);
        work = (uint8_t *)malloc((size_t)n + 16);
        if (fread(base, 1, (size_t)n, f) != (size_t)n)
            return 3;
        fclose(f);
        exercise(base, (size_t)n, base, (size_t)n);
        for (it = 0; it < rounds; it++) {
            size_t len = (size_t)n;
            unsigned mode = rnd() % 4u;
            memcpy(work, base, (size_t)n);
            if (mode == 0) {
                unsigned k, flips = 1 + rnd() % 4u;
                for (k = 0; k < flips; k++)
                    work[rnd() % (uint32_t)n] ^= (uint8_t)(1u << (rnd() % 8u));
            } else if (mode == 1) {
                len = rnd() % ((uint32_t)n + 1u);
            } else if (mode == 2) {
                work[rnd() % (uint32_t)(n < 96 ? n : 96)] = (uint8_t)rnd();
            } else {
                size_t i = rnd() % (uint32_t)n, j = rnd() % (uint32_t)n, k;
                for (k = 0; k < 8 && i + k < (size_t)n && j + k < (size_t)n; k++)
                    work[i + k] = base[j + k];
            }
            exercise(work, len, base, (size_t)n);
        }
        /* sequence files: write, reopen, read; then damaged copies of the file */
        if (seq_path) {
            HapSequenceWriter *w = NULL;
            HapSequenceReader *rd = NULL;
            if (HapSequenceWriterOpen(seq_path, 64, 64, 30, 1, &w) == HapResult_No_Error) {
                (void)HapSequenceWriterAppend(w, base, (unsigned long)n);
                (void)HapSequenceWriterAppend(w, base, (unsigned long)(n / 2 + 1));
                (void)HapSequenceWriterClose(w);
            }
            if (HapSequenceReaderOpen(seq_path, &rd) == HapResult_No_Error) {
                uint8_t *buf = (uint8_t *)malloc((size_t)n * 2 + 8);
                unsigned long offs[3];
                (void)HapSequenceReaderRead(rd, 0, 2, buf, (unsigned long)n * 2 + 8, offs);
                (void)HapSequenceReaderRead(rd, 1, 1, buf, 4, NULL);
                (void)HapSequenceReaderFrameBytes(rd, 5);
                free(buf);
                HapSequenceReaderClose(rd);
            }
            for (it = 0; it < 200; it++) {
                FILE *g = fopen(seq_path, "r+b");
                long size;
                if (!g)
                    break;
                fseek(g, 0, SEEK_END); size = ftell(g);
                fseek(g, (long)(rnd() % (uint32_t)size), SEEK_SET);
                fputc((int)(rnd() & 0xFF), g);
                fclose(g);
                if (HapSequenceReaderOpen(seq_path, &rd) == HapResult_No_Error) {
                    unsigned cnt = 0;
                    uint8_t *buf = (uint8_t *)malloc(1u << 16);
                    (void)HapSequenceReaderInfo(rd, NULL, NULL, NULL, NULL, &cnt);
                    if (cnt)
                        (void)HapSequenceReaderRead(rd, 0, 1, buf, 1u << 16, NULL);
                    free(buf);
                    HapSequenceReaderClose(rd);
                }
            }
            remove(seq_path);
        }
        free(base);
        free(work);
    }
    puts("ok");
    return 0;
}
