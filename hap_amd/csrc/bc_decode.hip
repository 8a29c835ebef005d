// bc_decode.hip -- DXT1 / DXT5 / scaled YCoCg-DXT5 (+ RGTC1 alpha plane) -> RGBA8 for gfx950.
//
// The reference stops at block-compressed texture bytes because its clients hand them to GPU
// texture units (README.md:4); CDNA has none, so a pipeline that wants pixels needs this kernel
// (SURVEY.md section 8f, rank 1).  Mirror image of bc_encode.hip: one 4x4 block per lane, 8/16-byte
// block load per lane (512 B / 1 KiB contiguous per wave), four 16-byte row stores per lane
// (1 KiB contiguous per wave-instruction).  Bounded by HBM: b read + 64 B written per block.
// Arithmetic follows oracle/bc_oracle.c (obc_decode_*) exactly; results are bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ int expand5(int q) { return (q << 3) | (q >> 2); }
__device__ __forceinline__ int expand6(int q) { return (q << 2) | (q >> 4); }
__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }

// byte `code` (0..7) of the 8-byte table hi:lo
__device__ __forceinline__ unsigned pick8(unsigned hi, unsigned lo, unsigned code)
{
    return __builtin_amdgcn_perm(hi, lo, code) & 0xFFu;
}

// 16 alpha-style values of an 8-byte block (S3TC alpha / RGTC1): the 8-entry palette is packed into
// two dwords and every pixel picks its byte with one v_perm_b32
__device__ __forceinline__ void decode_alpha(uint2 blk, int (&out)[16])
{
    const int a0 = (int)(blk.x & 255u), a1 = (int)((blk.x >> 8) & 255u);
    int v[8];
    v[0] = a0;
    v[1] = a1;
    if (a0 > a1) {
#pragma unroll
        for (int i = 1; i < 7; i++)
            v[i + 1] = (int)(__umul24((unsigned)((7 - i) * a0 + i * a1), 9363u) >> 16);     // / 7, exact below 13107
    } else {
#pragma unroll
        for (int i = 1; i < 5; i++)
            v[i + 1] = (int)(__umul24((unsigned)((5 - i) * a0 + i * a1), 13108u) >> 16);    // / 5, exact below 3277
        v[6] = 0;
        v[7] = 255;
    }
    const unsigned lo = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
    const unsigned hi = (unsigned)v[4] | ((unsigned)v[5] << 8) | ((unsigned)v[6] << 16) | ((unsigned)v[7] << 24);
    // 48 index bits = blk.x[16..31] | blk.y << 16 : pixels 0..7 in the low 24 bits, 8..15 above
    const unsigned lo24 = (blk.x >> 16) | ((blk.y & 0xFFu) << 16);
    const unsigned hi24 = blk.y >> 8;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[i] = (int)pick8(hi, lo, (lo24 >> (3 * i)) & 7u);
        out[8 + i] = (int)pick8(hi, lo, (hi24 >> (3 * i)) & 7u);
    }
}

// The same 16 values as packed pairs (pixel 2m in the low half, 2m + 1 in the high half of pairs[m]): one v_perm_b32
// fetches two palette bytes, the second selector byte of each half (0x0c) reads as zero
__device__ __forceinline__ void decode_alpha_pairs(uint2 blk, unsigned (&pairs)[8])
{
    const int a0 = (int)(blk.x & 255u), a1 = (int)((blk.x >> 8) & 255u);
    int v[8];
    v[0] = a0;
    v[1] = a1;
    if (a0 > a1) {
#pragma unroll
        for (int i = 1; i < 7; i++)
            v[i + 1] = (int)(__umul24((unsigned)((7 - i) * a0 + i * a1), 9363u) >> 16);
    } else {
#pragma unroll
        for (int i = 1; i < 5; i++)
            v[i + 1] = (int)(__umul24((unsigned)((5 - i) * a0 + i * a1), 13108u) >> 16);
        v[6] = 0;
        v[7] = 255;
    }
    const unsigned lo = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
    const unsigned hi = (unsigned)v[4] | ((unsigned)v[5] << 8) | ((unsigned)v[6] << 16) | ((unsigned)v[7] << 24);
    const unsigned lo24 = (blk.x >> 16) | ((blk.y & 0xFFu) << 16);
    const unsigned hi24 = blk.y >> 8;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const unsigned c_lo = (lo24 >> (6 * m)) & 63u, c_hi = (hi24 >> (6 * m)) & 63u;     // two 3-bit codes each
        pairs[m] = __builtin_amdgcn_perm(hi, lo, ((c_lo | (c_lo << 13)) & 0x00070007u) | 0x0c000c00u);
        pairs[4 + m] = __builtin_amdgcn_perm(hi, lo, ((c_hi | (c_hi << 13)) & 0x00070007u) | 0x0c000c00u);
    }
}

typedef short pk_i16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_i16 as_pk(unsigned v) { return __builtin_bit_cast(pk_i16, v); }
__device__ __forceinline__ unsigned as_u32(pk_i16 v) { return __builtin_bit_cast(unsigned, v); }

// per-channel palettes, one byte per entry: pal[c] = entry0 | entry1<<8 | entry2<<16 | entry3<<24
__device__ __forceinline__ void decode_palette(uint2 blk, bool dxt1_modes, unsigned (&pal)[3])
{
    const unsigned c0 = blk.x & 0xFFFFu, c1 = blk.x >> 16;
    const int e0[3] = {expand5(c0 >> 11), expand6((c0 >> 5) & 63), expand5(c0 & 31)};
    const int e1[3] = {expand5(c1 >> 11), expand6((c1 >> 5) & 63), expand5(c1 & 31)};
    const bool four = !dxt1_modes || c0 > c1;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // (/ 3 as one full-rate 24-bit multiply: exact below 32768)
        const int e2 = four ? (int)(__umul24((unsigned)(2 * e0[c] + e1[c]), 21846u) >> 16) : (e0[c] + e1[c]) / 2;
        const int e3 = four ? (int)(__umul24((unsigned)(e0[c] + 2 * e1[c]), 21846u) >> 16) : 0;
        pal[c] = (unsigned)e0[c] | ((unsigned)e1[c] << 8) | ((unsigned)e2 << 16) | ((unsigned)e3 << 24);
    }
}

// FMT: 0 DXT1, 1 DXT5, 2 YCoCg-DXT5; HAS_ALPHA: separate RGTC1 plane supplies A (Hap Q Alpha)
template <int FMT, bool HAS_ALPHA>
__device__ __forceinline__ void bc_decode_body(const uint8_t *__restrict__ blocks,
                                               const uint8_t *__restrict__ alpha_blocks,
                                               unsigned blocks_x, unsigned blocks_total,
                                               uint8_t *__restrict__ rgba, size_t row_bytes)
{
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    if (id >= blocks_total)
        return;
    const unsigned by = id / blocks_x, bx = id - by * blocks_x;
    int a[16];
    uint2 colour, luma_block = make_uint2(0u, 0u);
    if (FMT == 0) {
        colour = *reinterpret_cast<const uint2 *>(blocks + (size_t)id * 8u);
#pragma unroll
        for (int i = 0; i < 16; i++)
            a[i] = 255;
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(blocks + (size_t)id * 16u);
        if (FMT == 2)
            luma_block = make_uint2(v.x, v.y);          // YCoCg: luma, decoded in pairs below
        else
            decode_alpha(make_uint2(v.x, v.y), a);      // DXT5: alpha
        colour = make_uint2(v.z, v.w);
    }
    unsigned pal[3];
    decode_palette(colour, FMT == 0, pal);
    if (FMT == 2) {
        // undo the per-block chroma scaling once per palette entry (4x) instead of once per pixel:
        // pal[0]/pal[1] become (Co/s)+128 and (Cg/s)+128 (division truncating toward zero)
        unsigned co4 = 0, cg4 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = (int)(((pal[2] >> (8 * k)) & 255u) >> 3) + 1;               // 1..32
            int co = (int)((pal[0] >> (8 * k)) & 255u) - 128, cg = (int)((pal[1] >> (8 * k)) & 255u) - 128;
            // |x| / s for |x| <= 128 by a 16-bit reciprocal: floor(65536 / s) + 1 from v_rcp_f32 is exact here (the
            // quotient is an integer for powers of two, else at least 1/31 away from one), and so is the product's
            // top half for |x| < 516
            const unsigned m = (unsigned)(65536.0f * __builtin_amdgcn_rcpf((float)s)) + 1u;
            const int qo = (int)(__umul24((unsigned)abs(co), m) >> 16), qg = (int)(__umul24((unsigned)abs(cg), m) >> 16);
            co = co >= 0 ? qo : -qo;
            cg = cg >= 0 ? qg : -qg;
            co4 |= (unsigned)(co + 128) << (8 * k);
            cg4 |= (unsigned)(cg + 128) << (8 * k);
        }
        pal[0] = co4;
        pal[1] = cg4;
    }
    uint8_t *dst = rgba + (size_t)(4u * by) * row_bytes + 16u * (size_t)bx;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    if (FMT == 2) {
        // Two pixels per instruction (packed 16-bit): R = Y + (Co - Cg), G = Y + Cg, B = Y - Co - Cg with the three
        // offsets worked out once per palette entry (4) instead of once per pixel (16); a pixel pair fetches its two
        // entries' offsets with one v_perm_b32 per channel, adds the luma pair, clamps, and four v_perm_b32 interleave
        // R, G, B, A into two pixels.  (r04: 544 -> ~400 instructions per 64 blocks.)
        const pk_i16 k128 = {128, 128};
        const pk_i16 co01 = as_pk(__builtin_amdgcn_perm(0u, pal[0], 0x0c010c00u)) - k128, co23 = as_pk(__builtin_amdgcn_perm(0u, pal[0], 0x0c030c02u)) - k128;
        const pk_i16 cg01 = as_pk(__builtin_amdgcn_perm(0u, pal[1], 0x0c010c00u)) - k128, cg23 = as_pk(__builtin_amdgcn_perm(0u, pal[1], 0x0c030c02u)) - k128;
        const unsigned tr01 = as_u32(co01 - cg01), tr23 = as_u32(co23 - cg23);
        const unsigned tg01 = as_u32(cg01), tg23 = as_u32(cg23);
        const pk_i16 zero = {0, 0}, top = {255, 255};
        const unsigned tb01 = as_u32(zero - co01 - cg01), tb23 = as_u32(zero - co23 - cg23);
        unsigned ypair[8], apair[8];
        decode_alpha_pairs(make_uint2(luma_block.x, luma_block.y), ypair);
        if (HAS_ALPHA)
            decode_alpha_pairs(*reinterpret_cast<const uint2 *>(alpha_blocks + (size_t)id * 8u), apair);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            unsigned px[4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int m = 2 * r + h;
                const unsigned kk = (colour.y >> (4 * m)) & 15u;                        // two 2-bit palette indices
                // byte selectors of entries k0 (low half) and k1 (high half) of a table of four 16-bit values
                const unsigned sel = __umul24((kk | (kk << 14)) & 0x00030003u, 0x0202u) + 0x01000100u;
                const pk_i16 y2 = as_pk(ypair[m]);
                pk_i16 r2 = y2 + as_pk(__builtin_amdgcn_perm(tr23, tr01, sel));
                pk_i16 g2 = y2 + as_pk(__builtin_amdgcn_perm(tg23, tg01, sel));
                pk_i16 b2 = y2 + as_pk(__builtin_amdgcn_perm(tb23, tb01, sel));
                r2 = __builtin_elementwise_min(__builtin_elementwise_max(r2, zero), top);
                g2 = __builtin_elementwise_min(__builtin_elementwise_max(g2, zero), top);
                b2 = __builtin_elementwise_min(__builtin_elementwise_max(b2, zero), top);
                const unsigned rg = __builtin_amdgcn_perm(as_u32(g2), as_u32(r2), 0x06020400u);      // R0 G0 R1 G1
                const unsigned ba = __builtin_amdgcn_perm(HAS_ALPHA ? apair[m] : 0x00FF00FFu, as_u32(b2), 0x06020400u);   // B0 A0 B1 A1
                px[2 * h] = __builtin_amdgcn_perm(ba, rg, 0x05040100u);
                px[2 * h + 1] = __builtin_amdgcn_perm(ba, rg, 0x07060302u);
            }
            const v4u v = {px[0], px[1], px[2], px[3]};
            __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(dst + (size_t)r * row_bytes));
        }
        return;
    }
    int plane[16];
    if (HAS_ALPHA)
        decode_alpha(*reinterpret_cast<const uint2 *>(alpha_blocks + (size_t)id * 8u), plane);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned px[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int i = 4 * r + c;
            const unsigned k = (colour.y >> (2 * i)) & 3u;
            const int cr = (int)(__builtin_amdgcn_perm(0u, pal[0], k) & 0xFFu);
            const int cg = (int)(__builtin_amdgcn_perm(0u, pal[1], k) & 0xFFu);
            int R, G, B, A;
            if (FMT == 2) {
                const int co = cr - 128, cgg = cg - 128, y = a[i];
                R = clamp255(y + co - cgg);
                G = clamp255(y + cgg);
                B = clamp255(y - co - cgg);
                A = HAS_ALPHA ? plane[i] : 255;
            } else {
                R = cr;
                G = cg;
                B = (int)(__builtin_amdgcn_perm(0u, pal[2], k) & 0xFFu);
                A = HAS_ALPHA ? plane[i] : a[i];
            }
            px[c] = (unsigned)R | ((unsigned)G << 8) | ((unsigned)B << 16) | ((unsigned)A << 24);
        }
        {
            // streaming stores: the picture is written once and not read back by this kernel -- without the hint the
            // 16-byte stores of DXT1 / DXT5 run at 0.58 / 0.62 of HBM peak, with it at 0.76 (r04, 8K pictures)
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u v = {px[0], px[1], px[2], px[3]};
            __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(dst + (size_t)r * row_bytes));
        }
    }
}

template <int FMT, bool HAS_ALPHA>
__global__ __launch_bounds__(256) void bc_decode_kernel(const uint8_t *__restrict__ blocks,
                                                        const uint8_t *__restrict__ alpha_blocks,
                                                        unsigned blocks_x, unsigned blocks_total,
                                                        uint8_t *__restrict__ rgba, size_t row_bytes)
{
    bc_decode_body<FMT, HAS_ALPHA>(blocks, alpha_blocks, blocks_x, blocks_total, rgba, row_bytes);
}

// pictures of one geometry in one launch: picture blockIdx.z; table = [textures][alpha planes][pictures], `pictures`
// device addresses each; texture address 0 = not this launch's format: skip
template <int FMT, bool HAS_ALPHA>
__global__ __launch_bounds__(256) void bc_decode_batch_kernel(const uint64_t *__restrict__ table, unsigned pictures,
                                                              unsigned blocks_x, unsigned blocks_total, size_t row_bytes)
{
    const uint8_t *blocks = (const uint8_t *)table[blockIdx.z];
    if (!blocks)
        return;
    bc_decode_body<FMT, HAS_ALPHA>(blocks, (const uint8_t *)table[pictures + blockIdx.z], blocks_x, blocks_total,
                                   (uint8_t *)table[2u * pictures + blockIdx.z], row_bytes);
}

template <int FMT>
void launch_batch(const uint64_t *table, unsigned pictures, bool alpha, unsigned bx, unsigned by, size_t row_bytes, hipStream_t stream)
{
    const unsigned total = bx * by;
    const dim3 grid((total + 255u) / 256u, 1, pictures), block(256);
    if (alpha)
        hipLaunchKernelGGL((bc_decode_batch_kernel<FMT, true>), grid, block, 0, stream, table, pictures, bx, total, row_bytes);
    else
        hipLaunchKernelGGL((bc_decode_batch_kernel<FMT, false>), grid, block, 0, stream, table, pictures, bx, total, row_bytes);
}

template <int FMT>
void launch(const void *blocks, const void *alpha, unsigned bx, unsigned by, void *rgba, size_t row_bytes, hipStream_t stream)
{
    const unsigned total = bx * by;
    const dim3 grid((total + 255u) / 256u), block(256);
    if (alpha)
        hipLaunchKernelGGL((bc_decode_kernel<FMT, true>), grid, block, 0, stream, (const uint8_t *)blocks, (const uint8_t *)alpha, bx, total, (uint8_t *)rgba, row_bytes);
    else
        hipLaunchKernelGGL((bc_decode_kernel<FMT, false>), grid, block, 0, stream, (const uint8_t *)blocks, (const uint8_t *)nullptr, bx, total, (uint8_t *)rgba, row_bytes);
}

} // namespace

// format: HapTextureFormat of `blocks` (DXT1, DXT5, YCoCg-DXT5); alpha: optional RGTC1 plane.
// Returns 0 launched, 1 bad arguments.
extern "C" int hapgpu_launch_block_decode(const void *blocks, const void *alpha, unsigned width, unsigned height,
                                          unsigned format, void *rgba, size_t row_bytes, hipStream_t stream)
{
    if (!blocks || !rgba || width == 0 || height == 0 || (width & 3u) || (height & 3u) || row_bytes < (size_t)width * 4u)
        return 1;
    if (((uintptr_t)rgba | row_bytes) & 15u)
        return 1;
    if (((uintptr_t)blocks & (format == 0x83F0 ? 7u : 15u)) || ((uintptr_t)alpha & 7u))
        return 1;
    const unsigned bx = width / 4u, by = height / 4u;
    switch (format) {
    case 0x83F0: launch<0>(blocks, alpha, bx, by, rgba, row_bytes, stream); break;
    case 0x83F3: launch<1>(blocks, alpha, bx, by, rgba, row_bytes, stream); break;
    case 0x01: launch<2>(blocks, alpha, bx, by, rgba, row_bytes, stream); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// The same for `pictures` textures of one format and geometry: table (device memory) = texture addresses, alpha plane
// addresses (read when with_alpha), picture addresses, `pictures` of each; a texture address of 0 skips the picture.
// Alignment as above (the host checks it per picture).
extern "C" int hapgpu_launch_block_decode_batch(const uint64_t *table, unsigned pictures, int with_alpha, unsigned width,
                                                unsigned height, unsigned format, size_t row_bytes, hipStream_t stream)
{
    if (!table || pictures == 0 || pictures > 65535u || width == 0 || height == 0 || (width & 3u) || (height & 3u) ||
        row_bytes < (size_t)width * 4u || (row_bytes & 15u))
        return 1;
    const unsigned bx = width / 4u, by = height / 4u;
    switch (format) {
    case 0x83F0: launch_batch<0>(table, pictures, with_alpha != 0, bx, by, row_bytes, stream); break;
    case 0x83F3: launch_batch<1>(table, pictures, with_alpha != 0, bx, by, row_bytes, stream); break;
    case 0x01: launch_batch<2>(table, pictures, with_alpha != 0, bx, by, row_bytes, stream); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
