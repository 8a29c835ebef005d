/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the algorithms on the Hap hot path, used as the
 * checker for the HIP implementation in hap_amd/.  Nothing under hap_amd/ may
 * include, link or dlopen anything declared here; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() do.
 *
 * Three families:
 *   osnappy_*   Snappy block format (google/snappy, not vendored by the
 *               reference; call sites /root/reference/source/hap.c:313,453,
 *               612,813,890,899).  Pinned here against libsnappy 1.1.8:
 *               tests/test_oracle_pinning.py checks byte-identical
 *               compressed streams and identical decode results/status codes.
 *   ohap_*      Hap frame container + chunked second stage
 *               (/root/reference/source/hap.c:324-1188).  Pinned against the
 *               unmodified reference built into oracle/_ref/libhap_ref.so and
 *               against the golden byte dumps in tests/golden/.
 *   obc_*       RGBA -> DXT1 / DXT5 / scaled-YCoCg-DXT5 / RGTC1 block encoders.
 *               PARITY UNPINNED: the reference contains no block encoder
 *               (hap.h:82-104 takes pre-compressed texture bytes), so this
 *               oracle *defines* the integer algorithm the HIP kernels must
 *               reproduce bit-for-bit; block layouts follow the public S3TC /
 *               RGTC specs cited at documentation/HapVideoDRAFT.md:22-27.
 */
#ifndef HAP_ORACLE_H
#define HAP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Snappy ---------------------------------------------------------- */
enum { OSNAPPY_OK = 0, OSNAPPY_INVALID_INPUT = 1, OSNAPPY_BUFFER_TOO_SMALL = 2 };

size_t osnappy_max_compressed_length(size_t n);
int osnappy_compress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len);
int osnappy_uncompressed_length(const uint8_t *in, size_t n, size_t *result);
int osnappy_uncompress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len);

/* ---- Hap container ---------------------------------------------------- */
typedef void (*OHapWork)(void *p, unsigned index);
typedef void (*OHapCallback)(OHapWork fn, void *p, unsigned count, void *info);

unsigned long ohap_max_encoded_length(unsigned count, const unsigned long *lengths,
                                      const unsigned *formats, const unsigned *chunk_counts);
unsigned ohap_encode(unsigned count, const void **inputs, const unsigned long *input_bytes,
                     const unsigned *formats, const unsigned *compressors,
                     const unsigned *chunk_counts, void *out, unsigned long out_bytes,
                     unsigned long *out_used);
unsigned ohap_decode(const void *in, unsigned long in_bytes, unsigned index,
                     OHapCallback cb, void *info, void *out, unsigned long out_bytes,
                     unsigned long *out_used, unsigned *out_format);
unsigned ohap_texture_count(const void *in, unsigned long in_bytes, unsigned *count);
unsigned ohap_texture_format(const void *in, unsigned long in_bytes, unsigned index, unsigned *fmt);
unsigned ohap_texture_chunk_count(const void *in, unsigned long in_bytes, unsigned index, int *n);

/* ---- Block encoders ---------------------------------------------------- */
/* rgba: row-major RGBA8, row stride row_bytes; width/height multiples of 4.
 * out: blocks row-major, 8 B (DXT1, RGTC1) or 16 B (DXT5, YCoCg-DXT5). */
void obc_encode_dxt1(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out);
void obc_encode_dxt5(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out);
void obc_encode_ycocg_dxt5(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out);
void obc_encode_rgtc1_alpha(const uint8_t *rgba, unsigned w, unsigned h, size_t row_bytes, uint8_t *out);

/* Block decoders (for PSNR sanity and layout checks only). out = RGBA8. */
void obc_decode_dxt1(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *rgba);
void obc_decode_dxt5(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *rgba);
void obc_decode_ycocg_dxt5(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *rgba);
void obc_decode_rgtc1(const uint8_t *blocks, unsigned w, unsigned h, uint8_t *plane);

/* ---- Field-stream compressor (definition of the GPU's block-per-lane Snappy compressor) ---- */
/* One fragment (n <= 8192 bytes, whole blocks) of a block texture -> Snappy elements obeying the promises of the
 * private fragment table version 4; layout 4 = [2,6,4,4] (DXT5 / YCoCg-DXT5), 2 = [4,4] (DXT1), 6 = [2,6] (RGTC1).
 * Returns the bytes written to out (capacity >= n + n / 32 + 64); group_table[196] = 64 groups of
 * ceil(elements / 64) consecutive elements each, 24 bits per group little endian (compressed bytes | bytes produced
 * << 12), then the element count (LE16) and two zero bytes.  Not thread safe. */
unsigned ofs_compress_fragment(const uint8_t *src, unsigned n, unsigned layout, unsigned window_bytes, uint8_t *out,
                               uint8_t *group_table);
unsigned long ofs_texture_bytes(const uint8_t *tex, unsigned long bytes, unsigned chunks, unsigned layout);

#ifdef __cplusplus
}
#endif
#endif
