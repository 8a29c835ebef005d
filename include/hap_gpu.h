/*
 * hap_gpu.h -- device-resident, batched and RGBA entry points of hap_amd.
 *
 * hap.h (the reference API) takes one frame of already block-compressed
 * texture bytes at a time and has no RGBA input (reference hap.h:82-104).
 * The functions here cover what a GPU pipeline needs on top of that, without
 * touching hap.h:
 *
 *   - RGBA -> DXT1 / DXT5 / scaled YCoCg-DXT5 / RGTC1 block compression
 *     (the "external squish/DXT encoder" stage that reference clients run
 *     before HapEncode; absent from the reference tree),
 *   - whole batches of frames per call, all work enqueued on one HIP stream
 *     with a single host synchronisation at the end,
 *   - explicit contexts (device, stream, scratch) instead of the implicit
 *     per-process context hap.h uses.
 *
 * Plain C ABI: no HIP or C++ types appear in any signature.  Every pointer
 * argument documented as "host or device" is classified at run time
 * (hipPointerGetAttributes); device pointers are used in place, host pointers
 * are staged through pinned memory.
 *
 * Frames produced here are ordinary Hap frames: the reference decoder
 * (hap.c:993-1040) decodes them byte-identically.  With
 * HAPGPU_ENCODE_FRAGMENT_INDEX the Decode Instructions Container additionally
 * carries a private section (type 0x46) listing the compressed size of every
 * independently compressed Snappy fragment; decoders that do not know it skip
 * it (reference hap.c:701-703, HapVideoDRAFT.md:34), hap_amd's decoder uses it
 * to decode one chunk with many wavefronts.  The table also records the
 * granularity (1, 2 or 4 bytes) that every element of the streams honours.
 * The flag is off unless asked for -- also for plain hap.h HapEncode, where the
 * environment variable HAP_AMD_FRAGMENT_INDEX=1 stands in for it: whether every
 * OTHER parser of a frame skips unknown sections is not something this library
 * can know ("The private section").  Frames written without it
 * still consist of independent 8 KiB Snappy fragments; hap_amd's decoder finds
 * them with a scan (an element boundary at every 8 KiB of a chunk's output) and
 * decodes them one wavefront per fragment -- about a fifth of the speed the
 * table gives.
 *
 * Section 0x46, version 1:  [1][log2 F][granularity log2][match window / 256 B][LE32 compressed size x fragments]
 *               version 4:  [4][13][granularity log2 | fields per block << 4][window][LE32 size x fragments]
 *                           [196-byte group table x fragments]
 * Version 4 ("field streams": block textures, 8 KiB fragments) adds, per fragment, a table of 64 groups of its
 * elements -- the elements in stream order, ceil(N / 64) to a group, the last groups shorter or empty: 24 bits per
 * group, little endian, the group's compressed bytes | the bytes it produces << 12; then N (LE16) and two zero bytes --
 * and promises that no element crosses a 128-byte half-tile of output,
 * that every element starts and ends on a block field boundary (2 + 6 + 4 + 4, 4 + 4, 2 + 6, or -- layout 8, opaque
 * 16-byte blocks -- 4 + 4 + 4 + 4 bytes) and that every
 * copy offset is a whole number of blocks: the decoder's 64 lanes then each walk one group -- the same number of
 * elements, from a known input position to a known output position -- and produce one block per lane.  Every promise
 * is checked while decoding; a frame whose table lies is decoded again without it.  (Earlier builds wrote version 2,
 * one size byte per half-tile -- ignored: such frames decode like any other encoder's -- and version 3, 96-byte group
 * tables without the groups' output bytes: its fragment sizes are still used, one wavefront per fragment.)
 */
#ifndef HAP_AMD_HAP_GPU_H
#define HAP_AMD_HAP_GPU_H

#include "hap.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HapGpuContext HapGpuContext;

/* Encode flags */
#define HAPGPU_ENCODE_FRAGMENT_INDEX 0x1u   /* emit the private fragment-size section (type 0x46): for frames that only
                                               this library or hap.c-based readers will parse */
#define HAPGPU_ENCODE_COARSE_MATCHES 0x2u   /* size for speed, for the formats without a field layout of their own: BC7 /
                                               BC6H textures (opaque 16-byte blocks) go through the block kernels as four
                                               dwords per block, copy distances in whole blocks (table layout 8): about
                                               3.5x the encode and 3.8x the decode rate of the default position-per-lane
                                               streams for 0.41 instead of 0.38 of the texture size (8K, r04).  Other
                                               formats: Snappy elements on 32-bit boundaries (DXT1's are anyway; for
                                               DXT5 / YCoCg-DXT5 / RGTC1 the default field streams are both smaller
                                               and faster) */

#define HAPGPU_ENCODE_SMALLER_FILES 0x4u    /* smaller frames, slower: Snappy fragments of 64 KiB (matches up to 64 KiB back,
                                               as in libsnappy's blocks) with elements at any 16-bit position, and no
                                               private section.  8K YCoCg-DXT5: 0.360 of the texture size against 0.395
                                               (libsnappy: 0.336), compressing at a fifth of the default rate; such frames
                                               decode like another encoder's (block scan, about 300 GB/s of texture) */

#define HAPGPU_ENCODE_FINE_CHUNKS 0x8u      /* one second-stage chunk per Snappy fragment: the chunk counts of the call are
                                               replaced by HapGpuFineChunkCount() -- 8 KiB of texture per chunk where the
                                               texture divides that way (8K Hap Q: 4050 chunks).  The chunk tables every Hap
                                               parser reads (hap.c:265-300; 5 bytes per chunk) then list every independently
                                               compressed piece: nothing private in the frame, no scan when decoding.
                                               Size the output buffers with HapMaxEncodedLength() for THAT chunk count.
                                               8K Hap Q: +20 KB per frame (the private table: +810 KB); decoded one
                                               wavefront per chunk by the generic kernel ("Frames without a
                                               table").  May be combined with HAPGPU_ENCODE_FRAGMENT_INDEX. */

/* Decode flags */
#define HAPGPU_DECODE_IGNORE_FRAGMENT_INDEX 0x1u /* decode as a decoder unaware of section 0x46 would */
#define HAPGPU_DECODE_IGNORE_HALF_TILES 0x2u     /* use a version-4 table's fragment sizes only (the generic
                                                    fragment decoder), not its group tables: for A/B measurements */
#define HAPGPU_DECODE_NO_FIELD_GUESS 0x8u        /* frames whose chunks are each as short as one fragment (HAPGPU_ENCODE_FINE_CHUNKS)
                                                    and carry no private table: do not look for the block-per-lane decoder's
                                                    starting points (one lane per chunk walks its tags, then the chunk
                                                    decodes like a fragment with a table), decode every chunk with the
                                                    generic kernel -- what calls of fewer than 4096 such chunks do anyway.
                                                    The same for table-less frames whose chunks are many fragments long
                                                    (plain hap.h frames of this library): calls with 32768 or more 8 KiB
                                                    pieces found by the block scan run the pre-pass over the pieces */
#define HAPGPU_DECODE_GUESS_FIELDS 0x10u         /* ... do it however few the chunks / pieces are (tests) */
#define HAPGPU_DECODE_NO_BLOCK_SCAN 0x4u         /* decode other encoders' Snappy streams with one wavefront per
                                                    stream instead of looking for their 64 KiB blocks first: for A/B
                                                    measurements (environment HAP_AMD_NO_BLOCK_SCAN does the same) */

/* Creates a context on HIP device `device` (-1: the current device) with its
 * own non-blocking stream and growable scratch.  Returns a HapResult. */
unsigned int HapGpuCreate(int device, HapGpuContext **context);
void HapGpuDestroy(HapGpuContext *context);

/* The process-wide context used by the hap.h functions (created on first
 * use on the current device; HAP_AMD_DEVICE overrides). NULL if no GPU. */
HapGpuContext *HapGpuDefaultContext(void);

/* Log2 of the Snappy fragment size used by the compressor (10..16, default
 * 13 = 8 KiB).  Fragments are compressed independently of each other. */
unsigned int HapGpuSetFragmentLog2(HapGpuContext *context, unsigned int log2_bytes);

/* The chunk count HAPGPU_ENCODE_FINE_CHUNKS gives a texture: the smallest divisor of its block count (hap.c:277-300: every
 * chunk count must be one) that is at least bytes / 8 KiB rounded up -- chunks of at most 8 KiB, one Snappy fragment each;
 * where the block count has no divisor up to four times that number, the largest one below it (chunks of two fragments
 * and more: such frames decode like plain ones).  0 for arguments HapEncode would refuse.  Needs no GPU. */
unsigned int HapGpuFineChunkCount(unsigned long textureBytes, unsigned int textureFormat);

/* Blocks until everything enqueued on the context's stream has finished. */
unsigned int HapGpuSynchronize(HapGpuContext *context);

/* Number of frames this context has decoded a second time because their fragment table (section 0x46) did not
 * describe their streams, or because a stream split into 64 KiB blocks by the block scan turned out to copy across a
 * block start: table and scan are only ever hints, results are the same either way.  For tests and tools. */
unsigned long HapGpuTableFallbackCount(HapGpuContext *context);

/* Number of frames this context has encoded a second time: the block compressor writes a frame's compressed fragments
 * straight to their final places on the assumption that every chunk shrinks; a frame with a chunk that did not (stored
 * uncompressed, reference hap.c:460-466) is encoded again with the fragments gathered afterwards.  Same bytes either
 * way.  For tests and tools. */
unsigned long HapGpuPlacementRetryCount(HapGpuContext *context);

/* ... and how many of those were retried because a wavefront waited longer than its bound (about two milliseconds)
 * for the sizes of the fragments in front of its own: placing relies on the workgroups of a grid starting in index
 * order, which gfx950 does but HIP does not promise.  Such a frame switches placing off for the context's next 64 encode
 * calls, which gather (r06; until r05 for the rest of its life: one late neighbour under a profiler or on a shared GPU was a
 * permanent, nearly invisible change); 0 in every run so far.  For tests and tools. */
unsigned long HapGpuPlacementTimeoutCount(HapGpuContext *context);

/* Number of 64 KiB blocks of OTHER encoders' Snappy streams (frames without this library's private table, e.g. what the
 * reference's HapEncode writes: hap.c:448-476) that this context decoded with a whole workgroup each -- sixteen wavefronts
 * resolving the block's copies by pointer jumping -- instead of one wavefront walking its elements.  Which of the two
 * decodes a block changes nothing but the time it takes.  Waits for the context's stream.  For tests and tools. */
unsigned long HapGpuResolvedBlockCount(HapGpuContext *context);

/* RGBA8 (row-major, rowBytes stride, width/height multiples of 4) -> block
 * compressed texture.  textureFormat is one of RGB_DXT1, RGBA_DXT5,
 * YCoCg_DXT5, A_RGTC1 (A_RGTC1 compresses the alpha channel).  rgba and
 * output: host or device.  *outputBytesUsed = width/4 * height/4 * 8 or 16. */
unsigned int HapGpuCompressRGBA(HapGpuContext *context,
                                const void *rgba, unsigned int width, unsigned int height,
                                unsigned long rowBytes, unsigned int textureFormat,
                                void *output, unsigned long outputBytes,
                                unsigned long *outputBytesUsed);

/* Block-compressed texture -> RGBA8 (the stage a GPU's texture unit performs for the reference's
 * clients; CDNA has none).  textureFormat: RGB_DXT1, RGBA_DXT5 or YCoCg_DXT5 (converted back to
 * RGB); alphaTexture: optional A_RGTC1 plane that supplies A (Hap Q Alpha), else NULL / 0.
 * rgba must be 16-byte aligned with rowBytes a multiple of 16.  Host or device pointers. */
unsigned int HapGpuDecompressRGBA(HapGpuContext *context,
                                  const void *texture, unsigned long textureBytes, unsigned int textureFormat,
                                  const void *alphaTexture, unsigned long alphaBytes,
                                  unsigned int width, unsigned int height,
                                  void *rgba, unsigned long rowBytes);

/* Batched HapEncode: frame f is made of `count` textures
 * inputBuffers[f*count + i] of inputBuffersBytes[i] bytes each (every frame of
 * a batch has the same geometry).  Semantics, frame layout, chunk-count
 * limiting, store-raw decisions and result codes per frame follow HapEncode.
 * outputBuffers[f] must hold HapMaxEncodedLength() bytes.
 * results[f] receives the HapResult of frame f; the function result is the
 * first non-zero of them. */
unsigned int HapGpuEncodeFrames(HapGpuContext *context, unsigned int frameCount,
                                unsigned int count,
                                const void *const *inputBuffers,
                                const unsigned long *inputBuffersBytes,
                                const unsigned int *textureFormats,
                                const unsigned int *compressors,
                                const unsigned int *chunkCounts,
                                void *const *outputBuffers,
                                const unsigned long *outputBuffersBytes,
                                unsigned long *outputBuffersBytesUsed,
                                unsigned int *results,
                                unsigned int flags);

/* Batched RGBA -> Hap frame: block-compresses every frame into `count`
 * textures of textureFormats[] (e.g. {YCoCg_DXT5} for Hap Q, {YCoCg_DXT5,
 * A_RGTC1} for Hap Q Alpha, {RGB_DXT1} for Hap, {RGBA_DXT5} for Hap Alpha)
 * and packs them exactly as HapGpuEncodeFrames does; the intermediate
 * textures never leave HBM.  rgbaFrames[f]: host or device. */
unsigned int HapGpuEncodeFramesRGBA(HapGpuContext *context, unsigned int frameCount,
                                    const void *const *rgbaFrames,
                                    unsigned int width, unsigned int height,
                                    unsigned long rowBytes,
                                    unsigned int count,
                                    const unsigned int *textureFormats,
                                    const unsigned int *compressors,
                                    const unsigned int *chunkCounts,
                                    void *const *outputBuffers,
                                    const unsigned long *outputBuffersBytes,
                                    unsigned long *outputBuffersBytesUsed,
                                    unsigned int *results,
                                    unsigned int flags);

/* The same two calls in two halves, for pipelines: ...Begin checks the arguments, enqueues every launch of the call on
 * the context's stream and returns WITHOUT waiting for the GPU; HapGpuEncodeFramesFinish waits, fills
 * outputBuffersBytesUsed[] and results[] (the two arrays must live until then; every other argument array may go
 * once Begin has returned; the buffers themselves must of course stay) and encodes again whatever the first pass
 * could not place.  Between the halves the context takes no other call (Internal_Error) -- the client meanwhile
 * decodes the previous batch on ANOTHER context, parses, reads the next pictures: the encode kernels run under it
 * and the GPU never idles between calls (bench.py's step, DESIGN.md "Pipelined step").  Begin returns errors that are
 * known at once (then nothing is pending and Finish has nothing to do); Finish returns what the one-call form
 * returns.  At most 32768 frames per Begin.  A context destroyed between the halves waits for the launches and writes
 * no results.  The reference has no counterpart: HapEncode returns when its frame is
 * written (hap.h:98-104). */
unsigned int HapGpuEncodeFramesRGBABegin(HapGpuContext *context, unsigned int frameCount,
                                         const void *const *rgbaFrames,
                                         unsigned int width, unsigned int height,
                                         unsigned long rowBytes,
                                         unsigned int count,
                                         const unsigned int *textureFormats,
                                         const unsigned int *compressors,
                                         const unsigned int *chunkCounts,
                                         void *const *outputBuffers,
                                         const unsigned long *outputBuffersBytes,
                                         unsigned long *outputBuffersBytesUsed,
                                         unsigned int *results,
                                         unsigned int flags);
unsigned int HapGpuEncodeFramesBegin(HapGpuContext *context, unsigned int frameCount,
                                     unsigned int count,
                                     const void *const *inputBuffers,
                                     const unsigned long *inputBuffersBytes,
                                     const unsigned int *textureFormats,
                                     const unsigned int *compressors,
                                     const unsigned int *chunkCounts,
                                     void *const *outputBuffers,
                                     const unsigned long *outputBuffersBytes,
                                     unsigned long *outputBuffersBytesUsed,
                                     unsigned int *results,
                                     unsigned int flags);
unsigned int HapGpuEncodeFramesFinish(HapGpuContext *context);

/* Batched HapDecode of texture `index` of every frame.  No callback: all
 * chunks of all frames are decoded by the GPU.  Per-frame result codes,
 * bytes used and texture formats follow HapDecode (including the hardening
 * noted in INTEGRATION.md: out-of-range chunk tables are Bad_Frame instead of
 * an out-of-bounds read).  outputBytesUsed / outputTextureFormats may be NULL. */
unsigned int HapGpuDecodeFrames(HapGpuContext *context, unsigned int frameCount,
                                const void *const *inputBuffers,
                                const unsigned long *inputBuffersBytes,
                                unsigned int index,
                                void *const *outputBuffers,
                                const unsigned long *outputBuffersBytes,
                                unsigned long *outputBuffersBytesUsed,
                                unsigned int *outputTextureFormats,
                                unsigned int *results,
                                unsigned int flags);

/* The same for textures 0 .. textureCount-1 (1 or 2) of every frame in ONE batch: the arrays outputBuffers,
 * outputBuffersBytes, outputBuffersBytesUsed, outputTextureFormats and results have frameCount * textureCount
 * entries, entry f * textureCount + t belonging to texture t of frame f (what HapDecode(..., index = t, ...) would
 * be handed for that frame, reference hap.h:132-140).  A Hap Q Alpha stream decoded this way pays the per-call
 * costs (header read-back, launches, completion) once instead of once per texture. */
unsigned int HapGpuDecodeFrameTextures(HapGpuContext *context, unsigned int frameCount,
                                       const void *const *inputBuffers,
                                       const unsigned long *inputBuffersBytes,
                                       unsigned int textureCount,
                                       void *const *outputBuffers,
                                       const unsigned long *outputBuffersBytes,
                                       unsigned long *outputBuffersBytesUsed,
                                       unsigned int *outputTextureFormats,
                                       unsigned int *results,
                                       unsigned int flags);

/* Frames in, pixels out: every frame's second stage is undone (as HapGpuDecodeFrameTextures does, in one batch) and
 * its block texture expanded to an RGBA8 picture of width x height in rgbaFrames[f] (rowBytes: a multiple of 16,
 * at least width * 4; device pictures 16-byte aligned; host or device).  textureCount 1: the frames hold one
 * DXT1 / DXT5 / scaled-YCoCg-DXT5 texture (Hap, Hap Alpha, Hap Q; YCoCg is converted back to RGB the way the
 * reference's consumers do in their shader); textureCount 2: Hap Q Alpha frames, whose RGTC1
 * plane becomes the pictures' alpha.  The block textures live in the context's scratch only (at most 4 GiB of them at a
 * time: longer batches are worked through in slices).  results[f]:
 * HapDecode's code for the frame; Bad_Arguments for a frame whose texture is of another format or geometry than the
 * call says (BC7 / BC6H / lone RGTC1 textures have no pixel decoder here).  The reference has no counterpart: it
 * stops at the texture (hap.h:132-140) and leaves the pixels to the consumer's GPU. */
unsigned int HapGpuDecodeFramesRGBA(HapGpuContext *context, unsigned int frameCount,
                                    const void *const *inputBuffers,
                                    const unsigned long *inputBuffersBytes,
                                    unsigned int textureCount,
                                    void *const *rgbaFrames,
                                    unsigned int width, unsigned int height, unsigned long rowBytes,
                                    unsigned int *results,
                                    unsigned int flags);

/* --- one batch over several GPUs: independent frames per GPU ------------------- */

/* HapGpuEncodeFramesRGBA / HapGpuEncodeFrames / HapGpuDecodeFrames with the batch dealt out over `contextCount`
 * contexts -- normally one per GPU of the node (HapGpuCreate(device, ...)): frame f is worked on by context
 * f mod contextCount, each context on a host thread of its own, no data moves between devices and there is no
 * collective (frames are independent; a frame's buffers should live on the device that works on it, else they are
 * reached over the fabric or staged like any host pointer).  Arguments, per-frame results and the function result are
 * those of the single-context call, and so are the bytes written, whatever contextCount is (tests: 2, 3 and 8
 * contexts).  The contexts may share a device.  What the reference's clients do with a pool of threads calling
 * HapEncode / HapDecode frame by frame (hap.h:98-140). */
unsigned int HapGpuEncodeFramesRGBAOnDevices(HapGpuContext *const *contexts, unsigned int contextCount,
                                             unsigned int frameCount,
                                             const void *const *rgbaFrames,
                                             unsigned int width, unsigned int height, unsigned long rowBytes,
                                             unsigned int count,
                                             const unsigned int *textureFormats,
                                             const unsigned int *compressors,
                                             const unsigned int *chunkCounts,
                                             void *const *outputBuffers,
                                             const unsigned long *outputBuffersBytes,
                                             unsigned long *outputBuffersBytesUsed,
                                             unsigned int *results,
                                             unsigned int flags);
unsigned int HapGpuEncodeFramesOnDevices(HapGpuContext *const *contexts, unsigned int contextCount,
                                         unsigned int frameCount, unsigned int count,
                                         const void *const *inputBuffers,
                                         const unsigned long *inputBuffersBytes,
                                         const unsigned int *textureFormats,
                                         const unsigned int *compressors,
                                         const unsigned int *chunkCounts,
                                         void *const *outputBuffers,
                                         const unsigned long *outputBuffersBytes,
                                         unsigned long *outputBuffersBytesUsed,
                                         unsigned int *results,
                                         unsigned int flags);
unsigned int HapGpuDecodeFramesOnDevices(HapGpuContext *const *contexts, unsigned int contextCount,
                                         unsigned int frameCount,
                                         const void *const *inputBuffers,
                                         const unsigned long *inputBuffersBytes,
                                         unsigned int index,
                                         void *const *outputBuffers,
                                         const unsigned long *outputBuffersBytes,
                                         unsigned long *outputBuffersBytesUsed,
                                         unsigned int *outputTextureFormats,
                                         unsigned int *results,
                                         unsigned int flags);

/* --- one frame split over several GPUs by chunk groups ------------------------ */

/* HapDecode restricted to the chunks [firstChunk, firstChunk + chunkCount) of texture `index`:
 * outputBuffer is laid out as the WHOLE texture and only the group's byte range is written; the
 * rest is left untouched.  This is what a HapDecodeCallback that runs a subset of the work items
 * obtains from HapDecode (reference hap.h:113-130, hap.c:852-862); like there, frames that are
 * not chunked or have a single chunk are decoded completely.  *outputBufferBytesUsed is the
 * size of the whole texture.  inputBuffer / outputBuffer: host or device. */
unsigned int HapGpuDecodeChunkGroup(HapGpuContext *context,
                                    const void *inputBuffer, unsigned long inputBufferBytes,
                                    unsigned int index,
                                    unsigned int firstChunk, unsigned int chunkCount,
                                    void *outputBuffer, unsigned long outputBufferBytes,
                                    unsigned long *outputBufferBytesUsed,
                                    unsigned int *outputBufferTextureFormat);

/* Where each chunk of texture `index` lands in the decoded texture: the running sum of decoded
 * chunk sizes that the reference decoder builds (hap.c:794-838).  Writes chunkCount + 1 offsets
 * (the last one is the decoded size of the texture); Buffer_Too_Small if capacity is less.
 * Needs no GPU (inputBuffer: host or device). */
unsigned int HapGpuGetFrameTextureChunkLayout(const void *inputBuffer, unsigned long inputBufferBytes,
                                              unsigned int index, unsigned int capacity,
                                              unsigned long *decodedOffsets, unsigned int *chunkCount);

/* Joins frames that each hold one contiguous group of the chunks of the same texture(s) -- e.g.
 * bands of block rows encoded on different GPUs -- into one ordinary Hap frame whose chunk list
 * is the concatenation of the groups' lists, in the order given (frame layout: reference
 * hap.c:430-442, 562-598).  All groups must have the same texture count and formats.  A group
 * stored without chunks contributes a single chunk; if no chunk of a texture is compressed the
 * texture is written as a plain uncompressed section, as the reference does when Snappy gains
 * nothing (hap.c:478-495).  Fragment-size sections (type 0x46) are carried over when every
 * group has a compatible one.  Host pointers only; needs no GPU.
 * outputBufferBytes: the sum of the groups' sizes plus 64 always suffices. */
unsigned int HapGpuJoinChunkGroups(unsigned int groupCount,
                                   const void *const *groupFrames,
                                   const unsigned long *groupFramesBytes,
                                   void *outputBuffer, unsigned long outputBufferBytes,
                                   unsigned long *outputBufferBytesUsed);

/* The same join for group frames AND output in HIP device memory of `context`'s device (band frames that arrived
 * over xGMI): the groups' headers and tables are read through the host, the joined frame's headers are made there and
 * uploaded, every table and payload byte moves device to device.  Version-3 fragment tables (group tables) are
 * carried over by both joins when every group has one with the same block layout, so a joined frame decodes through
 * the block-per-lane kernel like its parts.  Bad_Arguments for host pointers. */
unsigned int HapGpuJoinChunkGroupsDevice(HapGpuContext *context, unsigned int groupCount,
                                         const void *const *groupFrames,
                                         const unsigned long *groupFramesBytes,
                                         void *outputBuffer, unsigned long outputBufferBytes,
                                         unsigned long *outputBufferBytesUsed);

/* --- measurement hooks (used by bench.py; see DESIGN.md "Measurement") --- */

/* Kernel classes whose launches are bracketed with HIP events on the
 * context's stream while profiling is enabled. */
enum HapGpuKernelClass {
    HapGpuKernel_BlockEncode = 0,
    HapGpuKernel_SnappyCompress = 1,
    HapGpuKernel_FramePack = 2,
    HapGpuKernel_FrameGather = 3,
    HapGpuKernel_DecodePlan = 4,
    HapGpuKernel_SnappyDecode = 5,
    HapGpuKernel_BlockDecode = 6,
    HapGpuKernel_BlockScan = 7,       /* finding the 64 KiB blocks of other encoders' Snappy streams */
    HapGpuKernel_EncodeFused = 8,     /* RGBA -> blocks -> Snappy fragments in one kernel (the calls that start from pictures) */
    HapGpuKernel_ClassCount = 9
};

/* enable != 0: record a start/stop event pair around every kernel launch. */
unsigned int HapGpuSetProfiling(HapGpuContext *context, unsigned int enable);
/* Drains recorded events: launches[k] and milliseconds[k] are ADDED to for every class k < classCount (the entries
 * the caller's arrays have; pass HapGpuKernel_ClassCount of the header the client was built with: a later library may
 * know more classes, and drops what the arrays have no room for). Synchronises. */
unsigned int HapGpuCollectProfileN(HapGpuContext *context, unsigned int classCount, unsigned long *launches, double *milliseconds);
/* The same for arrays of EIGHT entries (classes 0..7: the signature of the first release, which had no count). */
unsigned int HapGpuCollectProfile(HapGpuContext *context, unsigned long *launches, double *milliseconds);
/* Wall-clock bracket on the context's stream with HIP events. */
unsigned int HapGpuTimerStart(HapGpuContext *context);
unsigned int HapGpuTimerStop(HapGpuContext *context, double *milliseconds);

#ifdef __cplusplus
}
#endif

#endif /* HAP_AMD_HAP_GPU_H */
