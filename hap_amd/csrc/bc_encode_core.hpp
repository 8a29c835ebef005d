// bc_encode_core.hpp -- the block math of the RGBA8 -> DXT1 / DXT5 / scaled YCoCg-DXT5 / RGTC1 encoder: sixteen pixels of
// a lane in, the block's bits out.  Shared by bc_encode.hip (one block per lane, texture to memory) and the fused
// kernel of snappy_compress_blocks.hip (the block goes straight into the second stage).  The integer algorithm is the
// one defined by oracle/bc_oracle.c; results are bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hapbc {

constexpr int kFmtDXT1 = 0, kFmtDXT5 = 1, kFmtYCoCg = 2, kFmtRGTC1 = 3;
constexpr int kFmtYCoCgAlpha = 4;      // Hap Q Alpha: scaled YCoCg-DXT5 + RGTC1 alpha plane from one read of the RGBA

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
    const int a0 = hi, a1 = lo;            // (the exact range: every pixel lies on the ramp, no position needs clamping)
    unsigned lo24 = 0, hi24 = 0;       // 3-bit codes of pixels 0..7 and 8..15
    if (a0 != a1) {
        // oracle/bc_oracle.c: with d = a0 - a1 and u = a0 - a (0..d), the ramp position is
        // r = ((14 u + max(d - 6, 0)) * m) >> 20, m = floor(2^19 / d) + 1 -- the pixel's place on the ramp rounded to
        // the nearest of its 8 steps (x * m >> 20 = x / 2d), thresholds moved by the 3/7 the decoder's steps are
        // rounded down on average; 0..7 by construction.  Per pixel: one multiply-add, shift, code, insert.
        const int d = a0 - a1;
        // floor(2^19 / d): the reciprocal from v_rcp_f32, made exact
        unsigned q = (unsigned)(524288.0f * __builtin_amdgcn_rcpf((float)d));
        const int rem = 524288 - (int)__umul24(q, (unsigned)d);
        q += (rem >= d ? 1u : 0u) - (rem < 0 ? 1u : 0u);
        const unsigned m = q + 1u;
        // x = (14 (a0 - a) + bias) m as one multiply-add in a
        const int neg_m14 = -(int)(14u * m);                             // |.| < 2^23
        const int start = mad24(a0, (int)(14u * m), (int)__umul24((unsigned)max(d - 6, 0), m));
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned r = (unsigned)mad24(a[i], neg_m14, start) >> 20;
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

// 2-bit indices of 16 pixels for the palette p0, p1, (2 p0 + p1) / 3, (p0 + 2 p1) / 3 (oracle/bc_oracle.c,
// pick_indices): the entries lie at 3/3, 0/3, 2/3, 1/3 of the segment p1 .. p0, so the pixel is projected onto it,
//     t = (pixel - p1) . dir + len2 / 6 clamped to 0 .. len2 + len2 / 6,   dir = p0 - p1, len2 = |dir|^2,
//     pos = (t * floor(3 * 2^24 / len2)) >> 24,                            index = {1, 3, 2, 0}[pos].
// One unsigned byte dot product per pixel: channels whose direction is negative enter complemented (`flip` has 0xFF
// in those bytes, XOR-ed into the pixel unless FLIPPED says the caller did that already), the constants of the
// complement and of p1 . dir ride in the dot product's accumulator.  px, p0, p1: packed bytes, top byte zero.
struct projection {
    unsigned adir;      // |dir| per channel, packed
    unsigned flip;      // 0xFF where dir < 0
    int start;          // accumulator start: len2 / 6 - p1 . dir - 255 * (sum of |dir| over flipped channels)
    int top;            // len2 + len2 / 6
    unsigned m24;       // floor(3 * 2^24 / len2)
};

__device__ __forceinline__ projection make_projection(unsigned p0, unsigned p1)
{
    projection pr;
    // per-byte |p0 - p1| and the sign bytes: 9-bit lanes of a 32-bit subtraction would borrow across bytes, so per channel
    int dir[3];
    unsigned adir = 0, flip = 0;
    int neg = 0, base = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int a = (int)((p0 >> (8 * c)) & 255u), b = (int)((p1 >> (8 * c)) & 255u);
        dir[c] = a - b;
        const int ad = abs(dir[c]);
        adir |= (unsigned)ad << (8 * c);
        flip |= (dir[c] < 0 ? 0xFFu : 0u) << (8 * c);
        neg += dir[c] < 0 ? ad : 0;
        base = mad24(b, dir[c], base);
    }
    const unsigned len2 = __builtin_amdgcn_udot4(adir, adir, 0u, false);                 // 16 .. 195075
    const unsigned sixth = __umulhi(len2, 0xAAAAAAABu) >> 2;                             // len2 / 6
    // floor(3 * 2^24 / len2) from the float reciprocal, corrected with the integer remainder (off by one at most)
    unsigned m = (unsigned)(50331648.0f * __builtin_amdgcn_rcpf((float)len2));
    const int rem = (int)(50331648u - m * len2);
    m += (rem >= (int)len2 ? 1u : 0u) - (rem < 0 ? 1u : 0u);
    pr.adir = adir;
    pr.flip = flip;
    pr.start = (int)sixth - base - 255 * neg;
    pr.top = (int)(len2 + sixth);
    pr.m24 = m;
    return pr;
}

template <bool FLIPPED = false>
__device__ __forceinline__ unsigned project4(const unsigned (&px)[16], const projection &pr)
{
    unsigned pos2 = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const unsigned q = FLIPPED ? px[i] : px[i] ^ pr.flip;
        // (t is not clamped first: the product stays far inside 32 bits -- pixels lie within a few segment lengths of
        // p1 -- and clamping the position to 0..3 afterwards gives what the definition's clamp of t gives; the clamp is
        // left to the compiler's v_med3: an inline-asm reader right behind the dot product would miss its wait states)
        const int t = (int)__builtin_amdgcn_udot4(q, pr.adir, (unsigned)pr.start, false);
        const int pos = min(max(__mul24(t, (int)pr.m24) >> 24, 0), 3);
        // shift the two position bits in from the top: after 16 pixels pixel 0 sits in bits 1:0
        pos2 = __builtin_amdgcn_alignbit((unsigned)pos, pos2, 2);
    }
    // positions -> indices {1, 3, 2, 0}, all 16 at once: index high bit = pos.hi ^ pos.lo, low bit = ~pos.hi
    const unsigned hi = pos2 & 0xAAAAAAAAu;
    const unsigned idx = (hi ^ ((pos2 << 1) & 0xAAAAAAAAu)) | ((~hi >> 1) & 0x55555555u);
    return idx;
}

__device__ __forceinline__ unsigned pack3(int a, int b, int c) { return (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16); }

// the two end entries of the palette of a 5:6:5 endpoint pair, packed bytes (the blue field of a scaled YCoCg block
// carries the scale, not a colour: left out there)
__device__ __forceinline__ unsigned expand_565(unsigned c, bool blue)
{
    return pack3(expand5(c >> 11), expand6((c >> 5) & 63), blue ? expand5(c & 31) : 0);
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
    if (c0 != c1)
        idx = project4(px, make_projection(expand_565(c0, true), expand_565(c1, true)));
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
    // covariance sign: sum (2 Co - mo)(2 Cg - mg) = 4 sum Co Cg - 2 mg sum Co - 2 mo sum Cg + 16 mo mg with mo = lo + hi
    // of Co, mg of Cg: per pixel one product-accumulate and one packed add
    int prod = 0;
    pk_u16 sums = {0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int r;
        asm("v_mad_i32_i16 %0, %1, %1, %2 op_sel:[0,1,0,0]" : "=v"(r) : "v"(cc[i]), "v"(prod));   // low half x high half + prod
        prod = r;
        sums += __builtin_bit_cast(pk_u16, cc[i]);
    }
    const int mo = lo_o + hi_o, mg = lo_g + hi_g;
    const int cov = 4 * prod - 2 * mg * (int)sums.x - 2 * mo * (int)sums.y + 16 * mo * mg;
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
        const projection pr = make_projection(expand_565(c0, false), expand_565(c1, false));
        // project4 on the scaled pixels v = (c - 128) s + 128 without forming them, and without complementing: with the
        // direction SIGNED per channel (16-bit pair, one v_dot2 on the packed Co | Cg pair) the projection is
        // t = s (c . dir) + K, K = start + sum |dir| (128 - 128 s) over the channels that point up and
        // |dir| (127 + 128 s) over the ones that point down (project4's complement, multiplied out).  The dot product
        // starts from an offset that keeps it non-negative for the unsigned 24-bit multiply; the position is
        // (t m24) >> 24 = ((c . dir + offset) (s m24) + (K - s offset) m24) >> 24 in 32-bit wrap-around arithmetic (t m24
        // itself fits).
        const int a_o = (int)(pr.adir & 255u), a_g = (int)((pr.adir >> 8) & 255u);
        const bool down_o = (pr.flip & 0x00FFu) != 0u, down_g = (pr.flip & 0xFF00u) != 0u;
        const pk_i16 dir2 = {(short)(down_o ? -a_o : a_o), (short)(down_g ? -a_g : a_g)};
        const int K = pr.start + a_o * (down_o ? 127 + 128 * s : 128 - 128 * s) + a_g * (down_g ? 127 + 128 * s : 128 - 128 * s);
        constexpr int kDotOffset = 1 << 17;                                                  // > 2 x 255 x 255
        const unsigned sm = (unsigned)s * pr.m24;                                            // < 2^24
        const unsigned Km = (unsigned)(K - s * kDotOffset) * pr.m24;                         // mod 2^32
        unsigned pos2 = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            // (the offset from a scalar register: the compiler's choice, v_dot2c, accumulates into its destination and
            // pays a v_mov of the constant per pixel; the wait states a dot product needs before another VALU
            // instruction reads its result are part of the statement, the compiler does not see into it)
            unsigned dot;                                                                                                        // 1 .. 2^18
            asm("v_dot2_i32_i16 %0, %1, %2, %3\n\ts_nop 2" : "=v"(dot) : "v"(cc[i]), "v"(__builtin_bit_cast(unsigned, dir2)), "s"(kDotOffset));
            const int pos = min(max((int)(__umul24(dot, sm) + Km) >> 24, 0), 3);
            pos2 = __builtin_amdgcn_alignbit((unsigned)pos, pos2, 2);
        }
        const unsigned hi = pos2 & 0xAAAAAAAAu;
        idx = (hi ^ ((pos2 << 1) & 0xAAAAAAAAu)) | ((~hi >> 1) & 0x55555555u);
    }
    return make_uint2(c0 | (c1 << 16), idx);
}

// the 8 or 16 bytes of a block from its sixteen RGBA8 pixels (row-major); 8-byte formats leave z, w zero
template <int FMT>
__device__ __forceinline__ uint4 block_of(const unsigned (&p)[16])
{
    if (FMT == kFmtRGTC1) {
        int a[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            a[i] = (int)(p[i] >> 24);
        const uint2 ab = alpha_block(a);
        return make_uint4(ab.x, ab.y, 0u, 0u);
    } else if (FMT == kFmtDXT1) {
        unsigned px[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            px[i] = p[i] & 0x00FFFFFFu;
        const uint2 cb = colour_block(px);
        return make_uint4(cb.x, cb.y, 0u, 0u);
    } else if (FMT == kFmtDXT5) {
        int a[16];
        unsigned px[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            a[i] = (int)(p[i] >> 24);
            px[i] = p[i] & 0x00FFFFFFu;
        }
        const uint2 ab = alpha_block(a), cb = colour_block(px);
        return make_uint4(ab.x, ab.y, cb.x, cb.y);
    } else {
        int y[16];
        unsigned cc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            // Y = (R+2G+B+2)>>2 ; Co = ((R-B+1)>>1)+128 = (R+(255-B)+2)>>1 ; Cg = ((-R+2G-B+2)>>2)+128 =
            // ((255-R)+2G+(255-B)+4)>>2 -- three byte dot products (alpha weight 0); no clamp (oracle/bc_oracle.c)
            const unsigned q = p[i];
            y[i] = (int)(__builtin_amdgcn_udot4(q, 0x00010201u, 2u, false) >> 2);
            const unsigned co2 = __builtin_amdgcn_udot4(q ^ 0x00FF0000u, 0x00010001u, 2u, false);           // 2 Co: 2..512
            const unsigned cg4 = __builtin_amdgcn_udot4(q ^ 0x00FF00FFu, 0x00010201u, 4u, false);           // 4 Cg: 4..1024
            const pk_u16 raw = __builtin_bit_cast(pk_u16, co2 | (cg4 << 16)), sh = {1, 2};
            cc[i] = __builtin_bit_cast(unsigned, (pk_u16)(raw >> sh));                                       // halve / quarter, both at once: 1 .. 256
        }
        const uint2 ab = alpha_block(y), cb = ycocg_colour_block(cc);
        return make_uint4(ab.x, ab.y, cb.x, cb.y);
    }
}

} // namespace hapbc
