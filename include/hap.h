/*
 * hap.h -- public C API of hap_amd, the MI355X-native Hap frame codec.
 *
 * Drop-in for the Vidvox/hap reference header: the six functions, the three
 * enums and the two callback typedefs below have the same names, argument
 * order, types and numeric values as /root/reference/source/hap.h:40-152, so
 * a client built against the reference links against libhap_amd.so unchanged.
 * (Each declaration cites the reference lines it replaces.)
 *
 * What differs is where the work happens: the per-chunk Snappy second stage
 * (reference hap.c:448-476, 606-642, 885-904) runs as HIP kernels on gfx950.
 * Buffers may be ordinary host memory (staged over PCIe) or HIP device
 * memory (zero-copy); see include/hap_gpu.h for the device-resident, batched
 * and RGBA entry points that hap.h has no room for.
 */
#ifndef HAP_AMD_HAP_H
#define HAP_AMD_HAP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Texture formats: the OpenGL enumerants of S3TC / RGTC / BPTC, except scaled
 * YCoCg-DXT5 which has none and uses 0x01.  (reference hap.h:40-48) */
enum HapTextureFormat {
    HapTextureFormat_RGB_DXT1 = 0x83F0,
    HapTextureFormat_RGBA_DXT5 = 0x83F3,
    HapTextureFormat_YCoCg_DXT5 = 0x01,
    HapTextureFormat_A_RGTC1 = 0x8DBB,
    HapTextureFormat_RGBA_BPTC_UNORM = 0x8E8C,
    HapTextureFormat_RGB_BPTC_UNSIGNED_FLOAT = 0x8E8F,
    HapTextureFormat_RGB_BPTC_SIGNED_FLOAT = 0x8E8E
};

/* Second-stage compressor requested from HapEncode.  (reference hap.h:50-53) */
enum HapCompressor {
    HapCompressorNone = 0,
    HapCompressorSnappy = 1
};

/* Every function returning unsigned int returns one of these.
 * (reference hap.h:55-61) */
enum HapResult {
    HapResult_No_Error = 0,
    HapResult_Bad_Arguments = 1,
    HapResult_Buffer_Too_Small = 2,
    HapResult_Bad_Frame = 3,
    HapResult_Internal_Error = 4
};

/* Chunk fan-out contract of HapDecode.  (reference hap.h:66-67, 113-130)
 *
 * For a frame with more than one chunk HapDecode invokes `callback` exactly
 * once; the client must call function(p, i) for every i in [0, count) -- from
 * any threads, in any order -- and return only when all calls have returned.
 * In hap_amd function(p, i) is a cheap, thread-safe request marker: the
 * chunks are decoded by one kernel launch when the callback returns, so a
 * conforming client observes identical results while its worker threads stay
 * idle.  The callback is not invoked for single-chunk or unchunked frames. */
typedef void (*HapDecodeWorkFunction)(void *p, unsigned int index);
typedef void (*HapDecodeCallback)(HapDecodeWorkFunction function, void *p,
                                  unsigned int count, void *info);

/* Upper bound for the size of a frame made of `count` (1 or 2) textures, or
 * 0 for bad arguments.  Identical arithmetic to the reference (Snappy worst
 * case of every chunk + headers + tables).  (reference hap.h:76-79) */
unsigned long HapMaxEncodedLength(unsigned int count,
                                  unsigned long *lengths,
                                  unsigned int *textureFormats,
                                  unsigned int *chunkCounts);

/* Packs one or two block-compressed textures into a Hap frame.  The only
 * two-texture combination is YCoCg_DXT5 + A_RGTC1.  outputBufferBytes must
 * be at least HapMaxEncodedLength().  chunkCounts[i] is reduced to a divisor
 * of the texture's block count exactly as the reference does.
 * (reference hap.h:98-104) */
unsigned int HapEncode(unsigned int count,
                       const void **inputBuffers, unsigned long *inputBuffersBytes,
                       unsigned int *textureFormats,
                       unsigned int *compressors,
                       unsigned int *chunkCounts,
                       void *outputBuffer, unsigned long outputBufferBytes,
                       unsigned long *outputBufferBytesUsed);

/* Unpacks texture `index` (0 or 1) of a Hap frame.  callback must be non-NULL
 * even if it ends up unused; outputBufferBytesUsed may be NULL;
 * outputBufferTextureFormat must not be.  (reference hap.h:132-137) */
unsigned int HapDecode(const void *inputBuffer, unsigned long inputBufferBytes,
                       unsigned int index,
                       HapDecodeCallback callback, void *info,
                       void *outputBuffer, unsigned long outputBufferBytes,
                       unsigned long *outputBufferBytesUsed,
                       unsigned int *outputBufferTextureFormat);

/* Number of textures in a frame.  (reference hap.h:142) */
unsigned int HapGetFrameTextureCount(const void *inputBuffer, unsigned long inputBufferBytes,
                                     unsigned int *outputTextureCount);

/* Format of texture `index`.  (reference hap.h:147) */
unsigned int HapGetFrameTextureFormat(const void *inputBuffer, unsigned long inputBufferBytes,
                                      unsigned int index, unsigned int *outputBufferTextureFormat);

/* Chunk count of texture `index`.  (reference hap.h:152) */
unsigned int HapGetFrameTextureChunkCount(const void *inputBuffer, unsigned long inputBufferBytes,
                                          unsigned int index, int *chunk_count);

#ifdef __cplusplus
}
#endif

#endif /* HAP_AMD_HAP_H */
