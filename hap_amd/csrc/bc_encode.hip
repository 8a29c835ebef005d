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

namespace {

constexpr int kFmtDXT1 = 0, kFmtDXT5 = 1, kFmtYCoCg = 2, kFmtRGTC1 = 3;
constexpr int kFmtYCoCgAlpha = 4;      // Hap Q Alpha: scaled YCoCg-DXT5 + RGTC1 alpha plane from one read of the RGBA

// clamp(t, lo, hi) with lo <= hi: one v_med3_i32
__device__ __forceinline__ int imed3(int t, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "v"(lo), "v"(hi));
    return r;
}

// Inline-asm helpers below must never consume the result of a v_dot4 directly: gfx950 needs wait states between
// a dot product and a different VALU reader, and the compiler does not see through the asm to insert them.
// a * b + c on the 24-bit multiplier (full rate; the 32-bit one is quarter rate); |a|, |b| < 2^23
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int K>      // K: inline constant (-16..64)
__device__ __forceinline__ int mad24k(int a, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "i"(K), "v"(c));
    return r;
}

// exact floor(x / 7) for 0 <= x < 13107 and floor(x / 3) for 0 <= x < 32768 with one full-rate 24-bit multiply
__device__ __forceinline__ int div7(int x) { return (int)(__umul24((unsigned)x, 9363u) >> 16); }
__device__ __forceinline__ int div3(int x) { return (int)(__umul24((unsigned)x, 21846u) >> 16); }

__device__ __forceinline__ int quant5(int v) { int t = mad24k<31>(v, 128); return (t + (t >> 8)) >> 8; }
__device__ __forceinline__ int quant6(int v) { int t = mad24k<63>(v, 128); return (t + (t >> 8)) >> 8; }
__device__ __forceinline__ int expand5(int q) { return (q << 3) | (q >> 2); }
__device__ __forceinline__ int expand6(int q) { return (q << 2) | (q >> 4); }


// 8-byte alpha-style block: a0, a1, 16 x 3-bit codes (S3TC alpha / RGTC1 layout).
__device__ __forceinline__ uint2 alpha_block(const int (&a)[16])
{
    int lo = a[0], hi = a[0];
#pragma unroll
    for (int i = 1; i < 16; i++) {
        lo = min(lo, a[i]);
        hi = max(hi, a[i]);
    }
    const int inset = (hi - lo) >> 5;
    const int a0 = hi - inset, a1 = lo + inset;
    unsigned lo24 = 0, hi24 = 0;       // 3-bit codes of pixels 0..7 and 8..15
    if (a0 != a1) {
        // oracle/bc_oracle.c: ramp position r = #{ j < 7 : 2a < q_j + q_{j+1} } with q_j = floor(((7-j) a0 + j a1) / 7).
        // With u = a0 - a (clamped to 0..d, d = a0 - a1) and c_j = a0 - q_j = ceil(j d / 7) that is
        // r = #{ j : u > H_j }, H_j = floor((c_j + c_{j+1}) / 2) -- thresholds that grow with j.  Instead of testing all
        // seven, estimate from below, r_lo = floor(7 (u - 1) / d) (never more than one short: checked for every d, u),
        // and test the one threshold that decides: r = r_lo + (u > H[r_lo]), H_7 = 255.
        const int d = a0 - a1;
        int c[8];
        c[0] = 0;
#pragma unroll
        for (int j = 1; j < 7; j++)
            c[j] = div7(j * d + 6);
        c[7] = d;
        unsigned h_lo = 0, h_hi = 0xFF000000u;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const unsigned h = (unsigned)(c[j] + c[j + 1]) >> 1;
            if (j < 4)
                h_lo |= h << (8 * j);
            else
                h_hi |= h << (8 * (j - 4));
        }
        // floor(x / d) = (x * (floor(2^19 / d) + 1)) >> 19 for x <= 7 * 254; the reciprocal from v_rcp_f32, made exact
        unsigned q = (unsigned)(524288.0f * __builtin_amdgcn_rcpf((float)d));
        const int rem = 524288 - (int)__umul24(q, (unsigned)d);
        q += (rem >= d ? 1u : 0u) - (rem < 0 ? 1u : 0u);
        const unsigned r7 = 7u * (q + 1u);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int u = imed3(a0 - a[i], 0, d);
            const unsigned r_lo = __umul24(__builtin_elementwise_sub_sat((unsigned)u, 1u), r7) >> 19;   // (u - 1, not below 0)
            // H[r_lo]: one byte of the 8-byte table (selector bytes 1..3 = 0x0C give zero)
            const unsigned th = __builtin_amdgcn_perm(h_hi, h_lo, r_lo | 0x0C0C0C00u);
            const unsigned r = r_lo + ((unsigned)u > th ? 1u : 0u);
            // ramp position -> S3TC code: 0->0, 7->1, r->r+1 (byte table 00 02 03 04 | 05 06 07 01, one v_perm)
            const unsigned code = __builtin_amdgcn_perm(0x01070605u, 0x04030200u, r);
            // three bits in from the top: after 8 pixels the codes occupy bits 31:8, pixel 0 lowest
            if (i < 8)
                lo24 = __builtin_amdgcn_alignbit(code, lo24, 3);
            else
                hi24 = __builtin_amdgcn_alignbit(code, hi24, 3);
        }
        lo24 >>= 8;
        hi24 >>= 8;
    }
    const unsigned long long bits = (unsigned long long)lo24 | ((unsigned long long)hi24 << 24);
    const unsigned long long v = (unsigned long long)(unsigned)a0 | ((unsigned long long)(unsigned)a1 << 8) | (bits << 16);
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
}

// 2-bit index of the nearest of 4 palette entries for 16 pixels; px and pal are packed
// bytes (c0 | c1<<8 | c2<<16), top byte zero.  Lowest index wins ties.
template <bool COMPLEMENTED = false>      // COMPLEMENTED: px already holds 255 - p in its colour bytes
__device__ __forceinline__ unsigned nearest4(const unsigned (&px)[16], const unsigned (&pal)[4])
{
    // |p - c_k|^2 orders like |c_k|^2 - 2 p.c_k; scale by 4 and put k in the low bits so that one signed min
    // picks the smallest distance with the smallest index on ties: s_k = 4|c_k|^2 + k - 8 p.c_k.
    // With the complemented pixel p' = 255 - p (per byte), p.c_k = 255 sum(c_k) - p'.c_k, so
    // s_k = (4|c_k|^2 + k - 2040 sum(c_k)) + (p'.c_k << 3): one dot product and one shift-add per entry.
    int base[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        base[k] = 4 * (int)__builtin_amdgcn_udot4(pal[k], pal[k], 0u, false) + k -
                  2040 * (int)__builtin_amdgcn_udot4(pal[k], 0x00010101u, 0u, false);
    unsigned idx = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const unsigned q = COMPLEMENTED ? px[i] : px[i] ^ 0x00FFFFFFu;
        const int s0 = (int)(__builtin_amdgcn_udot4(q, pal[0], 0u, false) << 3) + base[0];
        const int s1 = (int)(__builtin_amdgcn_udot4(q, pal[1], 0u, false) << 3) + base[1];
        const int s2 = (int)(__builtin_amdgcn_udot4(q, pal[2], 0u, false) << 3) + base[2];
        const int s3 = (int)(__builtin_amdgcn_udot4(q, pal[3], 0u, false) << 3) + base[3];
        const int best = min(min(s0, s1), min(s2, s3));
        // shift the two index bits in from the top: after 16 pixels pixel 0 sits in bits 1:0
        idx = __builtin_amdgcn_alignbit((unsigned)best, idx, 2);
    }
    return idx;
}

__device__ __forceinline__ unsigned pack3(int a, int b, int c) { return (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16); }

__device__ __forceinline__ void palette_from_565(unsigned c0, unsigned c1, bool blue, unsigned (&pal)[4])
{
    const int r0 = expand5(c0 >> 11), g0 = expand6((c0 >> 5) & 63), b0 = blue ? expand5(c0 & 31) : 0;
    const int r1 = expand5(c1 >> 11), g1 = expand6((c1 >> 5) & 63), b1 = blue ? expand5(c1 & 31) : 0;
    pal[0] = pack3(r0, g0, b0);
    pal[1] = pack3(r1, g1, b1);
    pal[2] = pack3(div3(2 * r0 + r1), div3(2 * g0 + g1), div3(2 * b0 + b1));
    pal[3] = pack3(div3(r0 + 2 * r1), div3(g0 + 2 * g1), div3(b0 + 2 * b1));
}

// DXT1-style colour block from 16 packed RGB pixels (alpha byte already cleared).
__device__ __forceinline__ uint2 colour_block(const unsigned (&px)[16])
{
    int lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int v = (int)((px[i] >> (8 * c)) & 255u);
            lo[c] = min(lo[c], v);
            hi[c] = max(hi[c], v);
        }
    }
    int cov_rg = 0, cov_bg = 0;
    const int mr = lo[0] + hi[0], mg = lo[1] + hi[1], mb = lo[2] + hi[2];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int dr = 2 * (int)(px[i] & 255u) - mr;
        const int dg = 2 * (int)((px[i] >> 8) & 255u) - mg;
        const int db = 2 * (int)((px[i] >> 16) & 255u) - mb;
        cov_rg = mad24(dr, dg, cov_rg);
        cov_bg = mad24(db, dg, cov_bg);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int inset = (hi[c] - lo[c]) >> 4;
        lo[c] += inset;
        hi[c] -= inset;
    }
    const int ar = cov_rg < 0 ? lo[0] : hi[0], br = cov_rg < 0 ? hi[0] : lo[0];
    const int ab = cov_bg < 0 ? lo[2] : hi[2], bb = cov_bg < 0 ? hi[2] : lo[2];
    const unsigned qa = (unsigned)(quant5(ar) << 11 | quant6(hi[1]) << 5 | quant5(ab));
    const unsigned qb = (unsigned)(quant5(br) << 11 | quant6(lo[1]) << 5 | quant5(bb));
    const unsigned c0 = max(qa, qb), c1 = min(qa, qb);
    unsigned idx = 0;
    if (c0 != c1) {
        unsigned pal[4];
        palette_from_565(c0, c1, true, pal);
        idx = nearest4(px, pal);
    }
    return make_uint2(c0 | (c1 << 16), idx);
}

typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
typedef short pk_i16 __attribute__((ext_vector_type(2)));

// Colour half of a scaled YCoCg-DXT5 block; cc[i] = Co | Cg << 16, both biased by 128 (0..255): the box, the
// covariance terms and the scaling work on both halves at once (v_pk_min/max_u16, v_pk_mad_i16, v_mad_i32_i16).
__device__ __forceinline__ uint2 ycocg_colour_block(const unsigned (&cc)[16])
{
    pk_u16 lo = __builtin_bit_cast(pk_u16, cc[0]), hi = lo;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        const pk_u16 v = __builtin_bit_cast(pk_u16, cc[i]);
        lo = __builtin_elementwise_min(lo, v);
        hi = __builtin_elementwise_max(hi, v);
    }
    int lo_o = lo.x, hi_o = hi.x, lo_g = lo.y, hi_g = hi.y;
    const int m = max(max(128 - lo_o, hi_o - 128), max(128 - lo_g, hi_g - 128));
    const int s = m <= 31 ? 4 : (m <= 63 ? 2 : 1);
    int cov = 0;
    const pk_i16 mid = __builtin_bit_cast(pk_i16, lo + hi);            // (lo + hi per half: at most 510)
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const pk_i16 d = __builtin_bit_cast(pk_i16, cc[i]) * (short)2 - mid;  // 2 Co - (lo + hi) | 2 Cg - (lo + hi), |.| <= 255
        int r;
        asm("v_mad_i32_i16 %0, %1, %1, %2 op_sel:[0,1,0,0]" : "=v"(r) : "v"(d), "v"(cov));   // low half x high half + cov
        cov = r;
    }
    lo_o = (lo_o - 128) * s + 128; hi_o = (hi_o - 128) * s + 128;
    lo_g = (lo_g - 128) * s + 128; hi_g = (hi_g - 128) * s + 128;
    int ins = (hi_o - lo_o) >> 4; lo_o += ins; hi_o -= ins;
    ins = (hi_g - lo_g) >> 4; lo_g += ins; hi_g -= ins;
    const int ag = cov < 0 ? lo_g : hi_g, bg = cov < 0 ? hi_g : lo_g;
    const unsigned qa = (unsigned)(quant5(hi_o) << 11 | quant6(ag) << 5 | (s - 1));
    const unsigned qb = (unsigned)(quant5(lo_o) << 11 | quant6(bg) << 5 | (s - 1));
    const unsigned c0 = max(qa, qb), c1 = min(qa, qb);
    unsigned idx = 0;
    if (c0 != c1) {
        unsigned pal[4], px[16];
        palette_from_565(c0, c1, false, pal);
        // 255 - ((c - 128) s + 128) for both halves (16-bit arithmetic wraps to the right value) -- the index search
        // multiplies the complement --, then the two bytes side by side
        const pk_u16 scale = {(unsigned short)(0u - (unsigned)s), (unsigned short)(0u - (unsigned)s)};
        const unsigned short bias = (unsigned short)(127 + 128 * s);
        const pk_u16 off = {bias, bias};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned t = __builtin_bit_cast(unsigned, (pk_u16)(__builtin_bit_cast(pk_u16, cc[i]) * scale + off));
            px[i] = __builtin_amdgcn_perm(t, t, 0x0C0C0200u);            // bytes: t.0, t.2, zero, zero
        }
        idx = nearest4<true>(px, pal);
    }
    return make_uint2(c0 | (c1 << 16), idx);
}

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
    if (FMT == kFmtRGTC1) {
        int a[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            a[i] = (int)(p[i] >> 24);
        *reinterpret_cast<uint2 *>(out + id * 8u) = alpha_block(a);
    } else if (FMT == kFmtDXT1) {
        unsigned px[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            px[i] = p[i] & 0x00FFFFFFu;
        *reinterpret_cast<uint2 *>(out + id * 8u) = colour_block(px);
    } else if (FMT == kFmtDXT5) {
        int a[16];
        unsigned px[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            a[i] = (int)(p[i] >> 24);
            px[i] = p[i] & 0x00FFFFFFu;
        }
        const uint2 ab = alpha_block(a), cb = colour_block(px);
        *reinterpret_cast<uint4 *>(out + id * 16u) = make_uint4(ab.x, ab.y, cb.x, cb.y);
    } else {
        int y[16];
        unsigned cc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            // Y = (R+2G+B+2)>>2 ; Co = ((R-B+1)>>1)+128 = (R+(255-B)+2)>>1 ; Cg = ((-R+2G-B+2)>>2)+128 =
            // ((255-R)+2G+(255-B)+4)>>2 -- three byte dot products (alpha weight 0), upper clamp only
            const unsigned q = p[i];
            y[i] = (int)(__builtin_amdgcn_udot4(q, 0x00010201u, 2u, false) >> 2);
            const unsigned co = __builtin_amdgcn_udot4(q ^ 0x00FF0000u, 0x00010001u, 2u, false) >> 1;      // 1..256
            const unsigned cg = __builtin_amdgcn_udot4(q ^ 0x00FF00FFu, 0x00010201u, 4u, false) >> 2;      // 1..256
            const pk_u16 both = __builtin_bit_cast(pk_u16, co | (cg << 16)), top = {255, 255};
            cc[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(both, top));                      // upper clamp, both at once
        }
        const uint2 ab = alpha_block(y), cb = ycocg_colour_block(cc);
        *reinterpret_cast<uint4 *>(out + id * 16u) = make_uint4(ab.x, ab.y, cb.x, cb.y);
        if (FMT == kFmtYCoCgAlpha) {
            int a[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                a[i] = (int)(p[i] >> 24);
            *reinterpret_cast<uint2 *>(out2 + id * 8u) = alpha_block(a);
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
