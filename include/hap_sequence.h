/*
 * hap_sequence.h -- a minimal frame-sequence file and the disk -> GPU decode
 * pipeline around it ("stream/container adjacency").
 *
 * The reference library stops at single frames in memory: Hap frames normally
 * live in MOV/AVI files whose demuxing belongs to the host application
 * (reference README.md:10-32, HapVideoDRAFT.md:14).  This header is NOT a
 * replacement for those containers; it is the smallest file layout that lets a
 * batch pipeline be fed from storage:
 *
 *   offset 0   : 64-byte header
 *                  0  char[8]  "HAPSEQ1\0"
 *                  8  u32      version (1)
 *                  12 u32      width, 16 u32 height        (pixels; informative)
 *                  20 u32      rate numerator, 24 u32 rate denominator (frames per second; informative)
 *                  28 u32      frame count
 *                  32 u64      byte offset of the index
 *                  40 ..63     zero
 *   offset 64  : the frames, back to back, each exactly as HapEncode wrote it
 *   index      : (frame count + 1) x u64 byte offsets; frame i is [index[i], index[i+1])
 *
 * All integers little-endian.  Frame bytes are never interpreted here.
 * Results are HapResult codes (hap.h); I/O failures are HapResult_Internal_Error,
 * malformed files HapResult_Bad_Frame.
 */
#ifndef HAP_AMD_HAP_SEQUENCE_H
#define HAP_AMD_HAP_SEQUENCE_H

#include "hap_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HapSequenceWriter HapSequenceWriter;
typedef struct HapSequenceReader HapSequenceReader;

/* --- writing (no GPU involved) --- */
unsigned int HapSequenceWriterOpen(const char *path, unsigned int width, unsigned int height,
                                   unsigned int rateNumerator, unsigned int rateDenominator,
                                   HapSequenceWriter **writer);
/* frame: host memory holding one complete Hap frame */
unsigned int HapSequenceWriterAppend(HapSequenceWriter *writer, const void *frame, unsigned long frameBytes);
/* writes the index, patches the header, closes the file and frees the writer */
unsigned int HapSequenceWriterClose(HapSequenceWriter *writer);

/* --- reading (no GPU involved) --- */
unsigned int HapSequenceReaderOpen(const char *path, HapSequenceReader **reader);
void HapSequenceReaderClose(HapSequenceReader *reader);
unsigned int HapSequenceReaderInfo(const HapSequenceReader *reader, unsigned int *width, unsigned int *height,
                                   unsigned int *rateNumerator, unsigned int *rateDenominator,
                                   unsigned int *frameCount);
/* 0 when the index is out of range */
unsigned long HapSequenceReaderFrameBytes(const HapSequenceReader *reader, unsigned int frame);
/* Reads frames [first, first + count) back to back into buffer; offsets (count + 1 entries, may be
 * NULL) receives where each frame starts inside buffer.  Buffer_Too_Small if they do not fit. */
unsigned int HapSequenceReaderRead(HapSequenceReader *reader, unsigned int first, unsigned int count,
                                   void *buffer, unsigned long bufferBytes, unsigned long *offsets);

/* --- disk -> GPU pipeline ---
 * Decodes texture `index` of frames [first, first + count) of the file into outputBuffers[i]
 * (host or device, as for HapGpuDecodeFrames), `batch` frames per GPU submission (0: 16).  While the
 * GPU decodes one batch, a helper thread reads the next one from the file into the other of two
 * pinned host buffers, so that storage, PCIe upload and decode overlap.  Per-frame results as for
 * HapGpuDecodeFrames; the function result is the first non-zero of them (or an I/O error). */
unsigned int HapGpuDecodeSequence(HapGpuContext *context, HapSequenceReader *reader,
                                  unsigned int first, unsigned int count, unsigned int index,
                                  unsigned int batch,
                                  void *const *outputBuffers, const unsigned long *outputBuffersBytes,
                                  unsigned long *outputBuffersBytesUsed, unsigned int *outputTextureFormats,
                                  unsigned int *results);

/* --- GPU -> disk pipeline ---
 * Encodes `count` RGBA pictures (host or device memory, as for HapGpuEncodeFramesRGBA: same geometry, formats,
 * compressors and chunk counts for all) and appends the frames to the writer's file, `batch` pictures per GPU
 * submission (0: 16).  The GPU writes each batch's frames into one of two pinned host buffers; while it encodes the
 * next batch into the other, a helper thread appends the finished one to the file: block encode, Snappy stage, PCIe
 * download and storage overlap.  Per-frame results as for HapGpuEncodeFramesRGBA (NULL allowed); frames that fail
 * are not written and end the call.  frameBytes (NULL allowed): the size of every written frame. */
unsigned int HapGpuEncodeSequence(HapGpuContext *context, HapSequenceWriter *writer, unsigned int count,
                                  const void *const *rgbaFrames, unsigned int width, unsigned int height,
                                  unsigned long rowBytes, unsigned int textureCount, const unsigned int *textureFormats,
                                  const unsigned int *compressors, const unsigned int *chunkCounts,
                                  unsigned int flags, unsigned int batch, unsigned long *frameBytes,
                                  unsigned int *results);

#ifdef __cplusplus
}
#endif

#endif /* HAP_AMD_HAP_SEQUENCE_H */
