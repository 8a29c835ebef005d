// bc_encode.hip -- RGBA8 -> DXT1 / DXT5 / scaled YCoCg-DXT5 / RGTC1 block compression for gfx950.
//
// Replaces the "external squish/DXT encoder" stage that clients of the reference run in front
// of HapEncode (the reference itself has none: hap.h:82-104 takes compressed texture bytes).
// The integer algorithm is the one defined by oracle/bc_oracle.c; results are bit-identical.
//
// Mapping: one 4x4 block per lane.  Lane l of a wavefront loads the 16-byte pixel row segment
// of block bx = base + l for each of the block's 4 rows, so every load instruction of a wave
// covers 1 KiB of contiguous RGBA and every store 512 B / 1 KiB of contiguous blocks: fully
// coalesced without an LDS stage.  Bounded by HBM: 64 B read + 8/16 B written per block.
// Endpoint fitting uses per-lane min/max; index selection uses v_dot4_u32_u8 to get the
// |p-c|^2 ordering of four palette entries in four instructions per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bc_encode_core.hpp"

namespace {

using namespace hapbc;

template <int FMT, bool WIDE>
__device__ __forceinline__ void encode_block(const uint8_t *__restrict__ rgba, size_t row_bytes, unsigned blocks_x,
                                             uint8_t *__restrict__ out, uint8_t *__restrict__ out2 = nullptr)
{
    // one wavefront per 64 blocks of one block row: the row's address is scalar, no division per lane
    const unsigned by = blockIdx.y, bx = blockIdx.x * 64u + threadIdx.x;
    if (bx >= blocks_x)
        return;
    const size_t id = (size_t)by * blocks_x + bx;
    const uint8_t *src = rgba + (size_t)(4u * by) * row_bytes + 16u * bx;
    unsigned p[16];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (WIDE) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)r * row_bytes);
            p[4 * r + 0] = v.x; p[4 * r + 1] = v.y; p[4 * r + 2] = v.z; p[4 * r + 3] = v.w;
        } else {
            const unsigned *q = reinterpret_cast<const unsigned *>(src + (size_t)r * row_bytes);
            p[4 * r + 0] = q[0]; p[4 * r + 1] = q[1]; p[4 * r + 2] = q[2]; p[4 * r + 3] = q[3];
        }
    }
    if (FMT == kFmtRGTC1 || FMT == kFmtDXT1) {
        const uint4 b = block_of<FMT>(p);
        *reinterpret_cast<uint2 *>(out + id * 8u) = make_uint2(b.x, b.y);
    } else {
        *reinterpret_cast<uint4 *>(out + id * 16u) = block_of<FMT == kFmtYCoCgAlpha ? kFmtYCoCg : FMT>(p);
        if (FMT == kFmtYCoCgAlpha) {
            const uint4 a = block_of<kFmtRGTC1>(p);
            *reinterpret_cast<uint2 *>(out2 + id * 8u) = make_uint2(a.x, a.y);
        }
    }
}

template <int FMT, bool WIDE>
__global__ __launch_bounds__(64) void bc_encode_kernel(const uint8_t *__restrict__ rgba, size_t row_bytes,
                                                       unsigned blocks_x, unsigned blocks_total,
                                                       uint8_t *__restrict__ out)
{
    (void)blocks_total;
    encode_block<FMT, WIDE>(rgba, row_bytes, blocks_x, out);
}

// a batch of equally sized pictures in one launch: picture blockIdx.z, addresses from device arrays
template <int FMT, bool WIDE>
__global__ __launch_bounds__(64) void bc_encode_batch_kernel(const uint64_t *__restrict__ sources,
                                                             const uint64_t *__restrict__ outputs, size_t row_bytes,
                                                             unsigned blocks_x)
{
    const uint8_t *rgba = (const uint8_t *)sources[blockIdx.z];
    uint8_t *out = (uint8_t *)outputs[blockIdx.z];
    if (!rgba || !out)
        return;
    encode_block<FMT, WIDE>(rgba, row_bytes, blocks_x, out);
}

// Hap Q Alpha: both textures of a picture in one pass over its RGBA (64 + 16 + 8 bytes per block)
template <bool WIDE>
__global__ __launch_bounds__(64) void bc_encode_batch2_kernel(const uint64_t *__restrict__ sources,
                                                              const uint64_t *__restrict__ colour_outputs,
                                                              const uint64_t *__restrict__ alpha_outputs, size_t row_bytes,
                                                              unsigned blocks_x)
{
    const uint8_t *rgba = (const uint8_t *)sources[blockIdx.z];
    uint8_t *out = (uint8_t *)colour_outputs[blockIdx.z], *out2 = (uint8_t *)alpha_outputs[blockIdx.z];
    if (!rgba || !out || !out2)
        return;
    encode_block<kFmtYCoCgAlpha, WIDE>(rgba, row_bytes, blocks_x, out, out2);
}

template <int FMT>
void launch_batch(const uint64_t *sources, const uint64_t *outputs, unsigned pictures, size_t row_bytes, unsigned bx,
                  unsigned by, bool wide, hipStream_t stream)
{
    const dim3 grid((bx + 63u) / 64u, by, pictures), block(64);
    if (wide)
        hipLaunchKernelGGL((bc_encode_batch_kernel<FMT, true>), grid, block, 0, stream, sources, outputs, row_bytes, bx);
    else
        hipLaunchKernelGGL((bc_encode_batch_kernel<FMT, false>), grid, block, 0, stream, sources, outputs, row_bytes, bx);
}

template <int FMT>
void launch(const void *rgba, size_t row_bytes, unsigned bx, unsigned by, void *out, bool wide, hipStream_t stream)
{
    const unsigned total = bx * by;
    const dim3 grid((bx + 63u) / 64u, by), block(64);
    if (wide)
        hipLaunchKernelGGL((bc_encode_kernel<FMT, true>), grid, block, 0, stream, (const uint8_t *)rgba, row_bytes, bx, total, (uint8_t *)out);
    else
        hipLaunchKernelGGL((bc_encode_kernel<FMT, false>), grid, block, 0, stream, (const uint8_t *)rgba, row_bytes, bx, total, (uint8_t *)out);
}

} // namespace

// format: HapTextureFormat constant. Returns 0 when launched, 1 for bad arguments.
extern "C" int hapgpu_launch_block_encode(const void *rgba, unsigned width, unsigned height, size_t row_bytes,
                                          unsigned format, void *out, hipStream_t stream)
{
    if (!rgba || !out || width == 0 || height == 0 || (width & 3u) || (height & 3u) || row_bytes < (size_t)width * 4u)
        return 1;
    if (((uintptr_t)rgba & 3u) || (row_bytes & 3u))
        return 1;
    const unsigned bx = width / 4u, by = height / 4u;
    if ((unsigned long long)bx * by > 0xFFFFFFFFull / 256u * 255u)
        return 1;
    const bool wide = (((uintptr_t)rgba | row_bytes) & 15u) == 0;
    switch (format) {
    case 0x83F0: if ((uintptr_t)out & 7u) return 1; launch<kFmtDXT1>(rgba, row_bytes, bx, by, out, wide, stream); break;
    case 0x83F3: if ((uintptr_t)out & 15u) return 1; launch<kFmtDXT5>(rgba, row_bytes, bx, by, out, wide, stream); break;
    case 0x01: if ((uintptr_t)out & 15u) return 1; launch<kFmtYCoCg>(rgba, row_bytes, bx, by, out, wide, stream); break;
    case 0x8DBB: if ((uintptr_t)out & 7u) return 1; launch<kFmtRGTC1>(rgba, row_bytes, bx, by, out, wide, stream); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// Batch variant: `pictures` RGBA images of the same geometry whose addresses (and output addresses) are in device
// arrays; wide != 0 promises 16-byte aligned sources and row pitch.  Outputs must be 8/16-byte aligned.
extern "C" int hapgpu_launch_block_encode_batch(const uint64_t *sources, const uint64_t *outputs, unsigned pictures,
                                                unsigned width, unsigned height, size_t row_bytes, unsigned format,
                                                int wide, hipStream_t stream)
{
    if (!sources || !outputs || pictures == 0 || width == 0 || height == 0 || (width & 3u) || (height & 3u) ||
        row_bytes < (size_t)width * 4u || (row_bytes & 3u) || pictures > 65535u || height / 4u > 65535u)
        return 1;
    const unsigned bx = width / 4u, by = height / 4u;
    switch (format) {
    case 0x83F0: launch_batch<kFmtDXT1>(sources, outputs, pictures, row_bytes, bx, by, wide != 0, stream); break;
    case 0x83F3: launch_batch<kFmtDXT5>(sources, outputs, pictures, row_bytes, bx, by, wide != 0, stream); break;
    case 0x01: launch_batch<kFmtYCoCg>(sources, outputs, pictures, row_bytes, bx, by, wide != 0, stream); break;
    case 0x8DBB: launch_batch<kFmtRGTC1>(sources, outputs, pictures, row_bytes, bx, by, wide != 0, stream); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// Hap Q Alpha batch: scaled YCoCg-DXT5 to colour_outputs[i] and the RGTC1 alpha plane to alpha_outputs[i], one read of
// every RGBA picture.  Same argument rules as above.
extern "C" int hapgpu_launch_block_encode_batch_ycocg_alpha(const uint64_t *sources, const uint64_t *colour_outputs,
                                                            const uint64_t *alpha_outputs, unsigned pictures, unsigned width,
                                                            unsigned height, size_t row_bytes, int wide, hipStream_t stream)
{
    if (!sources || !colour_outputs || !alpha_outputs || pictures == 0 || width == 0 || height == 0 || (width & 3u) ||
        (height & 3u) || row_bytes < (size_t)width * 4u || (row_bytes & 3u) || pictures > 65535u || height / 4u > 65535u)
        return 1;
    const unsigned bx = width / 4u, by = height / 4u;
    const dim3 grid((bx + 63u) / 64u, by, pictures), block(64);
    if (wide)
        hipLaunchKernelGGL((bc_encode_batch2_kernel<true>), grid, block, 0, stream, sources, colour_outputs, alpha_outputs, row_bytes, bx);
    else
        hipLaunchKernelGGL((bc_encode_batch2_kernel<false>), grid, block, 0, stream, sources, colour_outputs, alpha_outputs, row_bytes, bx);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
