// snappy_compress_blocks.hip -- block-per-lane Snappy compressor for block textures ("field streams"), gfx950.
//
// Replaces the snappy_compress call-out of the reference's chunk loop (hap.c:448-476, call at hap.c:453) for DXT5 /
// YCoCg-DXT5 / DXT1 / RGTC1 textures.  The output is ordinary Snappy (literal / copy-1 / copy-2 elements: the
// reference decodes it unchanged) that keeps the promises of the private fragment table version 4
// (include/hap_gpu.h, snappy_decode_fields.hip): 8 KiB fragments, no element crosses a 128-byte half-tile, every
// element starts and ends on a block-field boundary, copy offsets are whole blocks; and a table of the bytes of 64
// groups of equally many elements per fragment, the decoder's starting points.  Its bytes are DEFINED by the
// scalar restatement oracle/field_stream_oracle.c; the tests compare the two byte for byte.
//
// One wavefront per fragment, a lane owns a 16-byte UNIT (one DXT5 block or two 8-byte blocks = 4 fields), so one
// wave-instruction covers 1 KiB ("step"); nothing but two small tables lives in LDS:
//
//   1. MATCH, lane = unit, 8 steps: the unit and the bytes 1..4 blocks in front of it come from memory as five
//      overlapping 16-byte loads per lane; 16 field comparisons per lane give a nibble per distance; index fields
//      also look up the most recent earlier block with the same value in a 512-entry table whose 64-bit entries
//      hold block number, field class and the full value (no second read to verify; ds_max_u64 inserts after the
//      step's lookups: the most recent block wins whatever the lane order).  A three-stage nibble transpose over
//      groups of 8 lanes (DPP + rotate + bit-field insert) turns "a nibble per distance per lane" into "a 32-bit
//      mask per distance per half-tile", one dword per lane to LDS.
//   2. CHOOSE, lane = half-tile (all 64 of the fragment at once), bit-parallel on 32-bit masks: copies are placed left
//      to right, nearest matching distance first, each running to the end of its match -- per round every uncovered
//      stretch starts one copy and an integer ADD carries it through its run of ones, so the loop runs as often as
//      the longest chain of touching copies (1.2 on average), not once per element.  Start, literal, "one more
//      byte" and distance masks, the half-tile's compressed size (popcounts) and -- one DPP scan -- its offset in
//      the fragment go back to LDS.
//   3. EMIT.  a: lane = unit, 8 steps: popcounts of the masks below the unit give its output offset; fields of
//      literal runs store their bytes at their final place.  b: lane = ELEMENT, 64 at a time in stream order (a fifth
//      of the fields start an element: a tag per field, predicated away, was half the kernel): phase 2 left a list
//      of (half-tile, field) per element; the half-tile's masks give stream offset, length and kind; one to three tag
//      bytes per element.
//   4. GROUP TABLE (version 4): the N elements of the fragment in 64 groups of ceil(N / 64); the first element of every
//      group leaves its stream offset and its output position behind during 3b; differences of neighbours are the
//      24-bit entries (compressed bytes | bytes produced << 12), N follows.
//
// HBM traffic: texture read once (the neighbour loads hit L1 / L2), compressed bytes written once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hapgpu_abi.h"
#include "measurement_guard.h"
#include "bc_encode_core.hpp"

namespace {

constexpr unsigned kFragBytes = 8192u;
constexpr unsigned kSteps = kFragBytes / 1024u;
constexpr unsigned kTableBits = 9u;                 // oracle/field_stream_oracle.c: OFS_TABLE_BITS
constexpr unsigned kDistances = 4u;

// Layout of a 16-byte unit: field k begins at fo(k) and has fs(k) bytes; copy distances are multiples of `block`.
//   4: DXT5 / YCoCg-DXT5 [2, 6, 4, 4], one block;  2: DXT1 [4, 4] x 2 blocks;  6: RGTC1 [2, 6] x 2 blocks;
//   8: an opaque 16-byte block [4, 4, 4, 4].
// (`code` is what the host puts into HapGpuTexEnc.reserved bits 16..19 for the layout.)
template <unsigned LAYOUT> struct unit_layout;
template <> struct unit_layout<4u> {
    static constexpr unsigned block = 16u, code = 4u;
    static constexpr unsigned small32 = 0x11111111u;     // 2-byte fields of a half-tile
    static constexpr unsigned big32 = 0x22222222u;       // 6-byte fields
    static constexpr unsigned four32 = 0xCCCCCCCCu;      // 4-byte fields
    static constexpr unsigned run3_12 = 0xBBBBBBBBu;     // starts whose 3-field run has >= 12 bytes
    static constexpr unsigned run15_61 = 0x22222222u;    // starts whose 15-field run has > 60 bytes
    __device__ static constexpr unsigned fo(unsigned k) { return k == 0u ? 0u : k == 1u ? 2u : k == 2u ? 8u : 12u; }
    __device__ static constexpr unsigned fs(unsigned k) { return k == 0u ? 2u : k == 1u ? 6u : 4u; }
    __device__ static constexpr unsigned cls(unsigned k) { return k == 1u ? 1u : k == 3u ? 3u : 0u; }
};
template <> struct unit_layout<2u> {
    static constexpr unsigned block = 8u, code = 10u;
    static constexpr unsigned small32 = 0u, big32 = 0u, four32 = 0xFFFFFFFFu, run3_12 = 0xFFFFFFFFu, run15_61 = 0u;
    __device__ static constexpr unsigned fo(unsigned k) { return 4u * k; }
    __device__ static constexpr unsigned fs(unsigned) { return 4u; }
    __device__ static constexpr unsigned cls(unsigned k) { return (k & 1u) ? 1u : 0u; }
};
template <> struct unit_layout<8u> {     // opaque 16-byte blocks (BC7, BC6H): four dwords, distances in whole blocks
    static constexpr unsigned block = 16u, code = 12u;
    static constexpr unsigned small32 = 0u, big32 = 0u, four32 = 0xFFFFFFFFu, run3_12 = 0xFFFFFFFFu, run15_61 = 0u;
    __device__ static constexpr unsigned fo(unsigned k) { return 4u * k; }
    __device__ static constexpr unsigned fs(unsigned) { return 4u; }
    __device__ static constexpr unsigned cls(unsigned k) { return k == 1u ? 1u : k == 3u ? 3u : 0u; }
};
template <> struct unit_layout<6u> {
    static constexpr unsigned block = 8u, code = 2u;
    static constexpr unsigned small32 = 0x55555555u, big32 = 0xAAAAAAAAu, four32 = 0u, run3_12 = 0xAAAAAAAAu,
                              run15_61 = 0xAAAAAAAAu;
    __device__ static constexpr unsigned fo(unsigned k) { return k == 0u ? 0u : k == 1u ? 2u : k == 2u ? 8u : 10u; }
    __device__ static constexpr unsigned fs(unsigned k) { return (k & 1u) ? 6u : 2u; }
    __device__ static constexpr unsigned cls(unsigned k) { return (k & 1u) ? 1u : 0u; }
};

// memory accesses at any byte address (the fragment of a client's texture may begin anywhere), in the global address
// space (pointers that come out of a descriptor as integers would otherwise be accessed with flat instructions,
// whose waits also stall on the LDS counter)
typedef const uint8_t __attribute__((address_space(1))) *gsrc_t;
typedef uint8_t __attribute__((address_space(1))) *gdst_t;
struct __attribute__((packed)) pk_u16 { uint16_t v; };
struct __attribute__((packed)) pk_u32 { uint32_t v; };
struct __attribute__((packed)) pk_u64 { uint32_t a, b; };
struct __attribute__((packed)) pk_u128 { uint32_t a, b, c, d; };
__device__ __forceinline__ void put8(gdst_t p, unsigned v) { *p = (uint8_t)v; }
__device__ __forceinline__ void put16(gdst_t p, unsigned v) { reinterpret_cast<pk_u16 __attribute__((address_space(1))) *>(p)->v = (uint16_t)v; }
__device__ __forceinline__ void put32(gdst_t p, unsigned v) { reinterpret_cast<pk_u32 __attribute__((address_space(1))) *>(p)->v = v; }
__device__ __forceinline__ void put64(gdst_t p, uint2 v)
{
    pk_u64 __attribute__((address_space(1))) *q = reinterpret_cast<pk_u64 __attribute__((address_space(1))) *>(p);
    q->a = v.x; q->b = v.y;
}
__device__ __forceinline__ void put128(gdst_t p, uint4 v)
{
    pk_u128 __attribute__((address_space(1))) *q = reinterpret_cast<pk_u128 __attribute__((address_space(1))) *>(p);
    q->a = v.x; q->b = v.y; q->c = v.z; q->d = v.w;
}
__device__ __forceinline__ uint4 get128(gsrc_t p)
{
    const pk_u128 __attribute__((address_space(1))) *q = reinterpret_cast<const pk_u128 __attribute__((address_space(1))) *>(p);
    return make_uint4(q->a, q->b, q->c, q->d);
}
__device__ __forceinline__ uint2 get64(gsrc_t p)
{
    const pk_u64 __attribute__((address_space(1))) *q = reinterpret_cast<const pk_u64 __attribute__((address_space(1))) *>(p);
    return make_uint2(q->a, q->b);
}

__device__ __forceinline__ unsigned rotr(unsigned v, unsigned n) { return __builtin_amdgcn_alignbit(v, v, n); }
__device__ __forceinline__ unsigned bfi(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ unsigned popc(unsigned v) { return (unsigned)__builtin_popcount(v); }
// 0 -> 0, anything else -> 1 (kept out of the compiler's hands: it turns min(x, 1) into a compare and a select
// through a scalar register pair, two instructions and wait states instead of one)
__device__ __forceinline__ unsigned nonzero(unsigned v)
{
    unsigned r;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(v));
    return r;
}
// all ones if bit `bit` of v is set, else 0
__device__ __forceinline__ unsigned bit_mask(unsigned v, unsigned bit) { return (unsigned)__builtin_amdgcn_sbfe((int)v, bit, 1u); }

__device__ __forceinline__ int scan_add(int v)          // inclusive, across the wavefront
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// value of lane ^ 4 / ^ 2 / ^ 1
__device__ __forceinline__ unsigned lane_xor4(unsigned v)
{
    int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);      // lanes 0-3, 8-11 of a row: from lane + 4
    t = __builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);          // lanes 4-7, 12-15: from lane - 4
    return (unsigned)t;
}
__device__ __forceinline__ unsigned lane_xor2(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); }
__device__ __forceinline__ unsigned lane_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }

// 1 where the field differs, per field of the unit, shifted to nibble `d`, OR-ed into acc
template <unsigned LAYOUT>
__device__ __forceinline__ unsigned differ_nibble(unsigned acc, const uint4 x, const uint4 y, unsigned d)
{
    const unsigned d0 = x.x ^ y.x, d1 = x.y ^ y.y, d2 = x.z ^ y.z, d3 = x.w ^ y.w;
    unsigned t0, t1, t2, t3;
    if (LAYOUT == 4u) {
        t0 = d0 & 0xFFFFu; t1 = (d0 & 0xFFFF0000u) | d1; t2 = d2; t3 = d3;
    } else if (LAYOUT == 2u || LAYOUT == 8u) {
        t0 = d0; t1 = d1; t2 = d2; t3 = d3;
    } else {
        t0 = d0 & 0xFFFFu; t1 = (d0 & 0xFFFF0000u) | d1; t2 = d2 & 0xFFFFu; t3 = (d2 & 0xFFFF0000u) | d3;
    }
    acc |= nonzero(t0) << (4u * d);
    acc |= nonzero(t1) << (4u * d + 1u);
    acc |= nonzero(t2) << (4u * d + 2u);
    acc |= nonzero(t3) << (4u * d + 3u);
    return acc;
}

// value of an index field: low 32 bits and the 16 bits above (0 for 4-byte fields)
template <unsigned LAYOUT>
__device__ __forceinline__ void index_field(const uint4 x, unsigned k, unsigned &lo, unsigned &hi)
{
    if (LAYOUT == 2u || LAYOUT == 8u) {
        lo = k == 1u ? x.y : x.w;
        hi = 0u;
    } else if (k == 1u) {
        lo = __builtin_amdgcn_alignbit(x.y, x.x, 16);
        hi = x.y >> 16;
    } else if (LAYOUT == 4u) {
        lo = x.w;
        hi = 0u;
    } else {
        lo = __builtin_amdgcn_alignbit(x.w, x.z, 16);
        hi = x.w >> 16;
    }
}

__device__ __forceinline__ unsigned table_slot(unsigned lo, unsigned hi, unsigned cls)
{
    const unsigned z = lo ^ rotr(hi, 19) ^ (cls << 29);
    return (z * 0x9E3779B1u) >> (32u - kTableBits);
}

// FUSED >= 0: the texture does not exist yet -- the wave makes its fragment's blocks from the frame's RGBA picture
// itself (bc_encode_core.hpp format FUSED), step by step, hands them to the match phase through a small LDS ring and
// leaves them at tex.src on the way (chunks that Snappy does not shrink are stored from there).  A wave is reading
// pixels while its neighbours on the SIMD are matching or emitting.
#ifndef SCB_MIN_WAVES
#define SCB_MIN_WAVES 1          // (measurement builds: waves per SIMD the register allocation must leave room for)
#endif
template <unsigned LAYOUT, int FUSED>
__global__ __launch_bounds__(64, SCB_MIN_WAVES) void snappy_compress_blocks_kernel(const HapGpuFrameEnc *__restrict__ frames,
                                                                    uint8_t *__restrict__ slots, unsigned slot_stride,
                                                                    uint32_t *__restrict__ frag_sizes,
                                                                    uint8_t *__restrict__ group_tables)
{
    using UL = unit_layout<LAYOUT>;
    constexpr unsigned B = UL::block;
    __shared__ unsigned long long table[1u << kTableBits];
    __shared__ __attribute__((aligned(16))) uint32_t masks[64u * 8u];
    __shared__ uint32_t bounds[66];                      // per group: stream offset of its first element | its output position << 16 (+ the end)

    const unsigned lane = threadIdx.x;
    // Single-texture launches take their frames interleaved: consecutive workgroups work on different frames, so the
    // fragments of one frame that are in flight together are few -- a placed stream waits for the sizes of everything
    // before it in its frame, and with a whole frame in flight every wavefront waited for the slowest of 4000.
#ifdef PLC_NO_INTERLEAVE
    const unsigned zf = blockIdx.z, x = blockIdx.x;
#else
    const unsigned linear = blockIdx.z * gridDim.x + blockIdx.x;
    const unsigned zf = gridDim.y == 1u ? linear % gridDim.z : blockIdx.z;
    const unsigned x = gridDim.y == 1u ? linear / gridDim.z : blockIdx.x;
#endif
    const HapGpuTexEnc tex = frames[zf].tex[blockIdx.y < 2u ? blockIdx.y : 0u];
    const unsigned tex_count = frames[zf].tex_count;
    if ((blockIdx.y >= tex_count) | (tex.compressor != 1u) | (((tex.reserved >> 16) & 0xFu) != UL::code) |
        (x >= tex.chunk_count * tex.frags_per_chunk) | (tex.src == 0) | (tex.chunk_bytes == 0) |
        (((tex.reserved >> 24) & 7u) != (unsigned)(FUSED + 1)))
        return;
    const unsigned chunk = x / tex.frags_per_chunk, fj = x - chunk * tex.frags_per_chunk;
    const unsigned begin = fj * kFragBytes;
    const unsigned n = min(kFragBytes, tex.chunk_bytes - begin);          // whole blocks (host-checked)
    const gsrc_t src = (gsrc_t)(tex.src + (uint64_t)chunk * tex.chunk_bytes + begin);
    const unsigned f = tex.frag_first + x;
    gdst_t out = (gdst_t)((uintptr_t)slots + (size_t)f * slot_stride);
    const bool placed = ((tex.reserved >> 27) & 1u) != 0u && blockIdx.y == 0u;
    const unsigned window = ((tex.reserved >> 8) & 0xFFu) ? ((tex.reserved >> 8) & 0xFFu) * 256u : 0xFFFFFFFFu;
    const bool want_sizes = ((tex.reserved >> 20) & 1u) != 0u && group_tables != nullptr;
    // where the fragment's group table goes: the table array (gathered into the frame later) -- or, for a placed
    // fragment, straight into the frame's fragment section
    gdst_t table_at = (gdst_t)((uintptr_t)group_tables + (size_t)f * HAP_GROUP_TABLE_BYTES);

    // table: empty
    {
        uint4 *t = reinterpret_cast<uint4 *>(table);
#pragma unroll
        for (unsigned i = 0; i < (sizeof(table) / 16u) / 64u; i++)
            t[i * 64u + lane] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // per-lane constants of the nibble transpose
    const unsigned keep4 = (lane & 4u) ? 0xFFFF0000u : 0x0000FFFFu;
    const unsigned keep2 = (lane & 2u) ? 0xFF00FF00u : 0x00FF00FFu;
    const unsigned keep1 = (lane & 1u) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
    const unsigned rot2 = (lane & 2u) ? 8u : 24u;
    const unsigned rot1 = (lane & 1u) ? 4u : 28u;
    // the last position a whole 16-byte (layouts of 8-byte blocks: 8-byte) load may start at: lanes beyond the data
    // load the last unit again (no lane-varying branch around the loads; what they compute is masked out)
    const unsigned last = n - (B == 16u ? 16u : 8u);

    // fused: a step of 64 units is made in kSub passes of 64 consecutive BLOCKS (one pass for 16-byte blocks, two for
    // 8-byte ones): the lane makes block 64 (kSub s + sub) + lane of the fragment from its four pixel rows (loaded
    // a pass ahead where registers allow) and puts it into an LDS ring, where the lane that owns the unit finds its blocks and their
    // neighbours.  A pass keeps sixteen pixel registers alive however many blocks a unit has.
    constexpr unsigned kSub = 16u / B;
    __shared__ __attribute__((aligned(16))) uint4 ring[FUSED >= 0 ? 68u : 1u];   // the step's units behind the last four of the one before
    const gsrc_t rgba = (gsrc_t)frames[zf].rgba;
    const unsigned row_bytes = frames[zf].rgba_row_bytes, blocks_x = frames[zf].rgba_blocks_x;
    unsigned bx[kSub], by[kSub], first_off = 0u;
    // Pixels a pass ahead of their use -- where that does not cost a wave: the YCoCg kernel has 88 registers without
    // the second pixel buffer (five waves per SIMD) and 112 with it (four), and five waves that wait for their pixels
    // beat four that do not (60 8K frames 2.78 -> 2.72 ms); the DXT5 kernel has four either way.
#ifdef SCB_PREFETCH
    constexpr bool kPrefetch = SCB_PREFETCH != 0;
#else
    constexpr bool kPrefetch = FUSED != hapbc::kFmtYCoCg;
#endif
    unsigned pix[kPrefetch ? 2 : 1][16];
    // pass t = kSub s + sub: its pixels
    auto load_pixels = [&](unsigned (&p)[16], unsigned t) {
        const unsigned sub = t % kSub;
        const bool there = (64u * t + lane + 1u) * B <= n;
        const unsigned at = there ? (4u * by[sub]) * row_bytes + 16u * bx[sub] : first_off;     // (no lane-varying branch around the loads)
#pragma unroll
        for (unsigned r = 0; r < 4u; r++) {
            const uint4 v = get128(rgba + (at + r * row_bytes));
            p[4u * r] = v.x; p[4u * r + 1u] = v.y; p[4u * r + 2u] = v.z; p[4u * r + 3u] = v.w;
        }
    };
    // the block of pass t + kSub from the block of pass t
    auto next_block = [&](unsigned sub) {
        bx[sub] += 64u * kSub;
        while (bx[sub] >= blocks_x) {
            bx[sub] -= blocks_x;
            by[sub] += 1u;
        }
    };
    if (FUSED >= 0) {
        const unsigned first_block = (unsigned)(((uint64_t)chunk * tex.chunk_bytes + begin) / B);
        {
            const unsigned fy = first_block / blocks_x, fx = first_block - fy * blocks_x;
            first_off = (4u * fy) * row_bytes + 16u * fx;
        }
#pragma unroll
        for (unsigned j = 0; j < kSub; j++) {
            const unsigned b = first_block + 64u * j + lane;
            by[j] = b / blocks_x;
            bx[j] = b - by[j] * blocks_x;
        }
        load_pixels(pix[0], 0u);
    }

    // (the fused kernel for 8-byte blocks has no registers left for the fragment's units between the match and the
    // emit walk: it reads them back from the texture it has just written)
    constexpr bool kKeepX = !(FUSED >= 0 && B == 8u);
    // ---- 1. match ----
    uint4 X[kKeepX ? kSteps : 1u];
    unsigned HD[kSteps];                 // table candidates of the unit's two index fields: distance in blocks, 0 = none
#pragma unroll
    for (unsigned s = 0; s < kSteps; s++) {
        const unsigned pos = (64u * s + lane) * 16u;
        if (kKeepX)
            X[s] = make_uint4(0, 0, 0, 0);
        HD[s] = 0u;
        if (64u * s * 16u >= n) {                  // (uniform) nothing left: half-tiles of this step read as empty
            masks[64u * s + lane] = 0u;
            continue;
        }
        const unsigned pc = min(pos, last);
        uint4 xs, Y[kDistances];
        if (FUSED >= 0) {
            uint4 made = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (unsigned sub = 0; sub < kSub; sub++) {
                const unsigned t = kSub * s + sub;
                if (!kPrefetch) {
                    if (t > 0u) {
                        if (t >= kSub)
                            next_block(t % kSub);
                        load_pixels(pix[0], t);
                    }
                } else if (t + 1u < kSub * kSteps && 64u * (t + 1u) * B < n) {
                    if (t + 1u >= kSub)
                        next_block((t + 1u) % kSub);
                    load_pixels(pix[(t + 1u) & 1u], t + 1u);
                }
                const uint4 blk = hapbc::block_of<FUSED>(pix[kPrefetch ? t & 1u : 0u]);
                made = blk;
                const unsigned bpos = (64u * t + lane) * B;
                // (the texture is kept for chunks that Snappy does not shrink -- they are stored from there.  A PLACED
                // call has no use for it: a frame with such a chunk is encoded again from its picture, without placing
                // (hap_batch.c), and r04's kernel wrote 3.7 x its algorithmic bytes for nobody)
                if (B == 16u) {
                    if (bpos + 16u <= n && !placed)
                        put128((gdst_t)(uintptr_t)(src + bpos), blk);
                    ring[4u + lane] = blk;
                } else {
                    if (bpos + 8u <= n)
                        put64((gdst_t)(uintptr_t)(src + bpos), make_uint2(blk.x, blk.y));
                    reinterpret_cast<uint2 *>(ring)[8u + 64u * sub + lane] = make_uint2(blk.x, blk.y);
                }
            }
            // the unit and the neighbours 1..4 blocks back: from the ring (a wave's LDS accesses complete in order)
            __syncthreads();
            const uint8_t *rb = reinterpret_cast<const uint8_t *>(ring) + (4u + lane) * 16u;
            xs = B == 16u ? made : *reinterpret_cast<const uint4 *>(rb);       // (a 16-byte block is the lane's own unit)
#pragma unroll
            for (unsigned d = 0; d < kDistances; d++) {
                if (B == 16u) {
                    Y[d] = *reinterpret_cast<const uint4 *>(rb - (d + 1u) * 16u);
                } else {
                    const uint2 ylo = *reinterpret_cast<const uint2 *>(rb - (d + 1u) * 8u), yhi = *reinterpret_cast<const uint2 *>(rb - (d + 1u) * 8u + 8u);
                    Y[d] = make_uint4(ylo.x, ylo.y, yhi.x, yhi.y);
                }
            }
            __syncthreads();
            if (lane >= 60u)
                ring[lane - 60u] = xs;
        } else if (B == 16u) {
            xs = get128(src + pc);
#pragma unroll
            for (unsigned d = 0; d < kDistances; d++) {
                const unsigned back = (d + 1u) * B;
                Y[d] = get128(src + (s == 0u ? max(pc, back) - back : pc - back));
            }
        } else {
            const uint2 lo = get64(src + pc), hi = get64(src + min(pos + 8u, last));
            xs = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
            for (unsigned d = 0; d < kDistances; d++) {
                const unsigned back = (d + 1u) * B;
                if (s == 0u) {                     // (the upper half of a unit may reach the fragment's first bytes)
                    const uint2 ylo = get64(src + max(pc, back) - back), yhi = get64(src + max(pc + 8u, back) - back);
                    Y[d] = make_uint4(ylo.x, ylo.y, yhi.x, yhi.y);
                } else {
                    Y[d] = get128(src + pc - back);
                }
            }
        }
        if (kKeepX)
            X[s] = xs;
        unsigned differ = 0;
#pragma unroll
        for (unsigned d = 0; d < kDistances; d++)
            differ = differ_nibble<LAYOUT>(differ, xs, Y[d], d);
        // fields that exist, and (first step) whose source d blocks back lies inside the fragment
        unsigned ok = pos + 16u <= n ? 0xFFFFu : 0u;
        if (B == 8u)
            ok = pos + 16u <= n ? 0xFFFFu : pos + 8u <= n ? 0x3333u : 0u;
        if (s == 0u) {
            if (B == 16u)
                ok &= (1u << (4u * min(lane, 4u))) - 1u;
            else
                ok &= lane == 0u ? 0x000Cu : lane == 1u ? 0x0CFFu : 0xFFFFu;
        }
        unsigned P = ~differ & ok;
        const unsigned eq_fixed = P;
        // table candidates of the index fields
        unsigned slot[2], klo[2], khi[2], valid[2];
        unsigned long long e[2];
#pragma unroll
        for (unsigned i = 0; i < 2u; i++) {
            const unsigned k = 2u * i + 1u;
            unsigned lo, hi;
            index_field<LAYOUT>(xs, k, lo, hi);
            klo[i] = lo;
            khi[i] = hi | (UL::cls(k) << 16);
            slot[i] = table_slot(lo, hi, UL::cls(k));
            valid[i] = (i == 0u ? pos + 8u <= n : pos + 16u <= n) ? 0xFFFFFFFFu : 0u;
            e[i] = table[slot[i]];
        }
        unsigned hd2[2];
#pragma unroll
        for (unsigned i = 0; i < 2u; i++) {
            const unsigned unit = 64u * s + lane;
            const unsigned blk = B == 16u ? unit : 2u * unit + i;
            const unsigned elo = (unsigned)e[i], ehi = (unsigned)(e[i] >> 32);
            const unsigned dist = blk - (ehi >> 18);
            const unsigned miss = (elo ^ klo[i]) | ((ehi & 0x3FFFFu) ^ khi[i]) | ~valid[i];
            const bool hit = (miss == 0u) & (dist * B <= window);
            hd2[i] = hit ? dist : 0u;
            P |= hit ? (dist * B >= 2048u ? 0x220000u : 0x020000u) << (2u * i) : 0u;
            // (units beyond the data insert a zero, which changes nothing; so do fields that repeat the same field one block
            // back: the first block of a run stays the candidate, and the lanes of a flat area do not queue up on one entry)
            const unsigned enter = valid[i] & ~(0u - ((eq_fixed >> (2u * i + 1u)) & 1u));
            atomicMax(&table[slot[i]], ((unsigned long long)((khi[i] | (blk << 18)) & enter) << 32) | (klo[i] & enter));
        }
        HD[s] = hd2[0] | (hd2[1] << 16);
        // nibbles of 8 lanes -> one 32-bit mask per kind: lane j of the group ends up with kind j of its half-tile
        P = bfi(keep4, P, rotr(lane_xor4(P), 16));
        P = bfi(keep2, P, rotr(lane_xor2(P), rot2));
        P = bfi(keep1, P, rotr(lane_xor1(P), rot1));
        masks[64u * s + lane] = P;
    }
    __syncthreads();

    // (the table has served: its first half now holds the candidates of every unit, for the lanes that will write the
    // elements; its second half, one byte per element in stream order: the field of its half-tile it starts at, | 0x80
    // for the first element of a half-tile -- written by phase 2)
    {
        uint32_t *hd_tab = reinterpret_cast<uint32_t *>(table);
#pragma unroll
        for (unsigned s = 0; s < kSteps; s++)
            hd_tab[64u * s + lane] = HD[s];
    }
    // ---- 2. choose: lane = half-tile ----
    unsigned counts_and_bytes = 0;          // inclusive scan over the half-tiles: elements << 16 | bytes (for the group table)
    {
        const uint4 ma = *reinterpret_cast<const uint4 *>(&masks[lane * 8u]);
        const uint4 mb = *reinterpret_cast<const uint4 *>(&masks[lane * 8u + 4u]);
        // valid fields of the half-tile (a prefix)
        const unsigned fields_total = (n >> 4) * 4u + ((n & 8u) ? 2u : 0u);
        const unsigned nv = fields_total > 32u * lane ? min(32u, fields_total - 32u * lane) : 0u;
        const unsigned valid = nv >= 32u ? 0xFFFFFFFFu : ((1u << nv) - 1u);
        unsigned E[kDistances] = {ma.x, ma.y, ma.z, ma.w}, A[kDistances] = {0u, 0u, 0u, 0u};
        unsigned U = 0;
#pragma unroll
        for (unsigned d = 0; d < kDistances; d++) {
            E[d] &= ~(UL::small32 & ~(E[d] >> 1) & ~(E[d] << 1));     // a 2-byte field only next to a neighbour that matches too
            U |= E[d];
        }
        unsigned front = U & ~(U << 1);
        while (__builtin_amdgcn_ballot_w64(front != 0u) != 0ull) {
            unsigned taken = 0, ends = 0;
#pragma unroll
            for (unsigned dd = 0; dd < kDistances; dd++) {
#ifdef HAP_BLK_FAR_FIRST
                const unsigned d = kDistances - 1u - dd;
#else
                const unsigned d = dd;
#endif
                const unsigned seeds = front & E[d] & ~taken;
                const unsigned sum = E[d] + seeds;
                A[d] |= (E[d] & ~sum) | seeds;
                ends |= sum & ~E[d];
                taken |= seeds;
            }
            front = ends & U;
        }
        const unsigned cov = A[0] | A[1] | A[2] | A[3];
        unsigned Hm = mb.x & ~cov;
        unsigned L = valid & ~(cov | Hm);
        if (UL::four32 != 0u) {
            // a copy of ONE 4-byte field with literal fields on both sides becomes literal bytes: one byte more, two
            // elements fewer (the decoder's lanes walk ceil(elements / 64) of them each; phase 3b a pass per 64)
            const unsigned cpy = cov | Hm;
            const unsigned lone4 = cpy & ~(cpy << 1) & ~(cpy >> 1) & UL::four32 & (L << 1) & (L >> 1);
            L |= lone4;
            Hm &= ~lone4;
#pragma unroll
            for (unsigned d = 0; d < kDistances; d++)
                A[d] &= ~lone4;
        }
        unsigned S = 0;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            S = Hm | (L & ~(L << 1));
#pragma unroll
            for (unsigned d = 0; d < kDistances; d++)
                S |= A[d] & ~(A[d] << 1);
            S &= valid;
            {
                // a copy is at most 64 bytes: the one that runs across field 16 (byte 64 of the half-tile) is cut there
                // if it is longer -- both pieces then fit.  a: its first field (the last start below 16; field 0 always
                // starts an element), b: the next start, or the end of the data
                constexpr unsigned FO = LAYOUT == 4u ? 0x0C080200u : (LAYOUT == 2u || LAYOUT == 8u) ? 0x0C080400u : 0x0A080200u;   // fo(k), a byte each
                const unsigned a = 31u - (unsigned)__builtin_clz((S & 0xFFFFu) | 1u);
                const unsigned b = 17u + (unsigned)__builtin_ctz(((S | ~valid) >> 17) | 0x8000u);
                const unsigned bytes = (b >> 2) * 16u + ((FO >> (8u * (b & 3u))) & 0xFFu) - (a >> 2) * 16u - ((FO >> (8u * (a & 3u))) & 0xFFu);
                const unsigned inside = (valid & ~S & ~L) >> 16;          // field 16 lies inside a copy
                S |= (inside & (bytes > 64u ? 1u : 0u)) << 16;
            }
            if (pass == 0 && UL::small32 != 0u) {
                // a copy never consists of a 2-byte field alone (no copy element is that short): such fields -- the tail
                // of a run that was cut, a match that nothing continues -- become literals, and the starts are found again
                const unsigned lone = UL::small32 & S & (((S | ~valid) >> 1) | 0x80000000u) & ~L & ~Hm & valid;
                L |= lone;
#pragma unroll
                for (unsigned d = 0; d < kDistances; d++)
                    A[d] &= ~lone;
            }
        }
        const unsigned Sx1 = ((S | ~valid) >> 1) | 0x80000000u;       // bit i: position i + 1 starts an element (or ends the data)
        const unsigned CS = S & ~L;
        const unsigned N1 = ~Sx1;
        // one more byte: copies of >= 12 bytes or from >= 2048 bytes back (copy-2), literal runs of > 60 bytes
        const unsigned C3 = (CS & N1 & (N1 >> 1) & ((N1 >> 2) | UL::run3_12)) | (Hm & mb.y);
        const unsigned Cn = L & ~S;
        const unsigned w2 = Cn & (Cn >> 1), w4 = w2 & (w2 >> 2), w8 = w4 & (w4 >> 4);
        const unsigned w12 = w8 & (w4 >> 8), w14 = w12 & (w2 >> 12), w15 = w14 & (Cn >> 14);
        const unsigned LS = S & L & ((w15 >> 1) | ((w14 >> 1) & UL::run15_61));
        const unsigned X3 = C3 | LS;
        const unsigned D0 = A[1] | A[3], D1 = A[2] | A[3];       // (distance - 1) of a copy's fields: the A[] are disjoint
        const unsigned lit_bytes = 4u * popc(L) - 2u * popc(L & UL::small32) + 2u * popc(L & UL::big32);
        const unsigned total = lit_bytes + popc(S) + popc(CS) + popc(X3);
        // one scan for the bytes (low half: at most 64 x 144) and the elements (high half: at most 2048) before the half-tile
        const unsigned count = popc(S);
        const unsigned both = (unsigned)scan_add((int)(total | (count << 16)));
        const unsigned incl = both & 0xFFFFu;
        const unsigned base = incl - total;
        *reinterpret_cast<uint4 *>(&masks[lane * 8u]) = make_uint4(S, L, X3, D0);
        *reinterpret_cast<uint4 *>(&masks[lane * 8u + 4u]) = make_uint4(D1, Hm, Sx1, base | (((both >> 16) - count) << 16));
        if (lane == 63u) {
            if (placed) {
                // the size is out before the bytes are: whoever comes later in the frame can place itself.  (Relaxed,
                // device scope: nobody reads anything else this wave wrote; an acquire / release pair at this scope
                // writes back and invalidates a whole L2 every time -- 12 times the kernel's duration)
                __hip_atomic_store(&frag_sizes[f], incl | HAPGPU_FRAG_PUBLISHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                frag_sizes[f] = incl;
            }
        }
        counts_and_bytes = both;
        // the elements in stream order, for the lanes of phase 3b: element (first ordinal of the half-tile + j) is the
        // j-th set bit of S -- a loop over the busiest half-tile's elements, five instructions a turn (finding the r-th
        // set bit per element in phase 3b instead cost 45 instructions per element pass)
        {
            uint8_t *elfield = reinterpret_cast<uint8_t *>(table) + 2048u;
            unsigned left = S, at = (both >> 16) - count, first = 0x80u;
            while (__builtin_amdgcn_ballot_w64(left != 0u) != 0ull) {
                if (left != 0u) {
                    elfield[at] = (uint8_t)(first | (unsigned)__builtin_ctz(left));
                    left &= left - 1u;
                    at += 1u;
                    first = 0u;
                }
            }
        }
    }
    __syncthreads();

    // ---- placed streams: where the fragment's bytes go in the frame ----
    // payload of the texture's section + a varint per chunk up to this one + the bytes of every fragment before this
    // one: whole chunks from their accumulators (complete when they have counted all their fragments), the fragments
    // of this chunk from their published sizes.  Everything waited for belongs to workgroups with a LOWER index.
    // Forward progress therefore rests on an assumption HIP does not state: that the workgroups of a grid are
    // dispatched in index order, so that a waiting wavefront never holds the slot its predecessor needs.  gfx950 (and
    // every GCN / CDNA part so far) dispatches that way, this library is built for gfx950 only, and the wait is
    // BOUNDED anyway: kPlacedMaxPolls polls of ~0.4 us -- two milliseconds, a hundred times what a neighbour takes to
    // publish -- then the wavefront gives up, writes to its slot and marks the frame; the host encodes a marked frame
    // again through slots, counts it (HapGpuPlacementTimeoutCount) and stops placing for the context's lifetime: a
    // neighbour kernel on another stream, CU masking or a serialising profiler cost one late call, not a cliff per call.
    constexpr unsigned kPlacedMaxPolls = 1u << 12;
    if (placed) {
        const unsigned long long *acc = reinterpret_cast<const unsigned long long *>(frames[zf].chunk_acc);
        const uint32_t *mine = frag_sizes + tex.frag_first + chunk * tex.frags_per_chunk;
        unsigned long long sum = 0;
        unsigned in_chunk = 0;
        unsigned tries = 0;
        bool settled = true;
        // (group after group of 64 words, oldest first, each polled on its own until it is complete: what is waited for
        // is almost always the nearest neighbours only.  These loads are served by the memory side and their number is
        // what counts: fetching five groups at once "to save latency" made the kernel 17 % slower)
        for (unsigned c0 = 0; c0 < chunk && settled; c0 += 64u) {
            const unsigned c = c0 + lane;
            for (;;) {
                const unsigned long long v = c < chunk ? __hip_atomic_load(&acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                       : (1ull << 32);
                if (__builtin_amdgcn_ballot_w64((unsigned)(v >> 32) == 0u) == 0ull) {
                    sum += v & 0xFFFFFFFFull;
                    settled = __builtin_amdgcn_ballot_w64((unsigned)(v >> 32) != 1u) == 0ull;      // (2: a chunk that gave up)
                    break;
                }
                if (++tries > kPlacedMaxPolls) {
                    settled = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        for (unsigned k0 = 0; k0 < fj && settled; k0 += 64u) {
            const unsigned k = k0 + lane;
            for (;;) {
                const unsigned v = k < fj ? __hip_atomic_load(&mine[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : HAPGPU_FRAG_PUBLISHED;
                if (__builtin_amdgcn_ballot_w64((v & HAPGPU_FRAG_PUBLISHED) == 0u) == 0ull) {
                    in_chunk += v & ~HAPGPU_FRAG_PUBLISHED;
                    break;
                }
                if (++tries > kPlacedMaxPolls) {
                    settled = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        const unsigned own_bytes = (unsigned)__builtin_amdgcn_readlane((int)counts_and_bytes, 63) & 0xFFFFu;
        if (settled) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                sum += __shfl_xor(sum, d);
                in_chunk += __shfl_xor(in_chunk, d);
            }
            // the chunk's last fragment has just added up the whole chunk: its total, for the chunks behind
            // (plain stores and loads only: read-modify-writes from every wavefront on a few words were served one
            // after the other, 60 ns each)
            if (fj + 1u == tex.frags_per_chunk && lane == 0u)
                __hip_atomic_store(const_cast<unsigned long long *>(&acc[chunk]), (1ull << 32) | (in_chunk + own_bytes), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            sum += in_chunk;
            const unsigned n_chunks = tex.chunk_count, fpc = tex.frags_per_chunk;
            const bool with_tiles = tex.emit_index && want_sizes;
            const unsigned index_len = tex.emit_index ? 8u + (with_tiles ? 4u + HAP_GROUP_TABLE_BYTES : 4u) * n_chunks * fpc : 0u;
            const unsigned cb = tex.chunk_bytes;
            const unsigned vlen = cb < (1u << 7) ? 1u : cb < (1u << 14) ? 2u : cb < (1u << 21) ? 3u : cb < (1u << 28) ? 4u : 5u;
            const unsigned long long at = frames[zf].dst + frames[zf].outer_header_len + tex.header_len + 4u +
                                          (5u * n_chunks + 8u + index_len) + (unsigned long long)vlen * (chunk + 1u) + sum;
            const unsigned at_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)at);
            const unsigned at_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(at >> 32));
            out = (gdst_t)(uintptr_t)(((unsigned long long)at_hi << 32) | at_lo);
            // (frame_pack.hip: compressor table behind the section's three headers, the sizes, the fragment section's
            // header and its four bytes, a size per fragment, then the group tables in fragment order)
            if (with_tiles)
                table_at = (gdst_t)(uintptr_t)(frames[zf].dst + frames[zf].outer_header_len + tex.header_len + 8u + n_chunks + 4u +
                                               4u * n_chunks + 8u + 4u * n_chunks * fpc +
                                               (unsigned long long)(chunk * fpc + fj) * HAP_GROUP_TABLE_BYTES);
        } else if (lane == 0u) {
            // (a frame with such a fragment is encoded again through slots, and the chunks behind it need not wait
            // for the total; bit 0 of the frame's reserved word tells the host that it was a timeout)
            __hip_atomic_fetch_or(const_cast<uint32_t *>(&frames[zf].reserved), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fj + 1u == tex.frags_per_chunk)
                __hip_atomic_store(const_cast<unsigned long long *>(&acc[chunk]), 2ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- 3a. emit the literal bytes: lane = unit ----
    // popcounts of the masks below the unit give its output offset; a field that is part of a literal run stores its
    // bytes behind the headers and bytes of what comes before it (only the stores are predicated)
    const unsigned j4 = 4u * (lane & 7u);
    const unsigned below = (1u << j4) - 1u;
#pragma unroll
    for (unsigned s = 0; s < kSteps; s++) {
        if (64u * s * 16u >= n)
            continue;
        const unsigned hh = 8u * s + (lane >> 3);
        const uint4 ma = *reinterpret_cast<const uint4 *>(&masks[hh * 8u]);
        const unsigned S = ma.x, L = ma.y, X3 = ma.z, CS = S & ~L;
        unsigned off = (masks[hh * 8u + 7u] & 0xFFFFu) + 4u * popc(L & below) + 2u * popc(L & UL::big32 & below) -
                       2u * popc(L & UL::small32 & below) + popc(S & below) + popc(CS & below) + popc(X3 & below);
        const unsigned sj = S >> j4, lj = L >> j4, xj = X3 >> j4;
        uint4 xs;
        if (kKeepX) {
            xs = X[s];
        } else {
            const unsigned pos = (64u * s + lane) * 16u;
            const uint2 lo = get64(src + min(pos, last)), hi = get64(src + min(pos + 8u, last));
            xs = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        constexpr bool kDwords = LAYOUT == 2u || LAYOUT == 8u;
        const unsigned fw[4] = {xs.x, kDwords ? xs.y : xs.x >> 16, xs.z, LAYOUT == 4u ? xs.w : kDwords ? xs.w : xs.z >> 16};
#pragma unroll
        for (unsigned k = 0; k < 4u; k++) {
            const unsigned is_s = bit_mask(sj, k), is_l = bit_mask(lj, k), x3 = bit_mask(xj, k);
            // element bytes in front of the field's own: literal header 1 (+1), copy 2 (+1)
            off += is_s & ((is_l & 1u) + (~is_l & 2u) + (x3 & 1u));
            if (is_l) {
                if (UL::fs(k) == 2u) {
                    put16(out + off, fw[k]);
                } else if (UL::fs(k) == 4u) {
                    put32(out + off, fw[k]);
                } else {
                    put16(out + off, fw[k]);
                    put32(out + off + 2, k == 1u ? xs.y : xs.w);
                }
            }
            off += is_l & UL::fs(k);
        }
    }

    // ---- 3b. emit the elements' tags: lane = element, 64 at a time in stream order ----
    // The field of element e comes from the list phase 2 left, its half-tile from a count of the first-of-half-tile
    // marks up to it; the half-tile's masks then give its
    // stream offset (the bytes of what lies below), its length (fields up to the next start) and its kind.  The elements whose
    // ordinal is a multiple of G = ceil(N / 64) begin the groups of the fragment table (version 4): their offsets and output positions go to
    // bounds[].
    {
        const unsigned both = counts_and_bytes;
        const unsigned stream_bytes = (unsigned)__builtin_amdgcn_readlane((int)both, 63) & 0xFFFFu;
        const unsigned elements = (unsigned)__builtin_amdgcn_readlane((int)both, 63) >> 16;        // >= 1
        const unsigned G = (elements + 63u) >> 6;
        const unsigned inv = (1u << 20) / G + 1u;                              // x / G = (x inv) >> 20 for x < 2^11 + 32
        const uint32_t *hd_tab = reinterpret_cast<const uint32_t *>(table);
        {
            bounds[lane] = stream_bytes | (n << 16);                           // (groups beyond the last element: empty)
            if (lane < 2u)
                bounds[64u + lane] = stream_bytes | (n << 16);
        }
        __syncthreads();
        constexpr unsigned FO = LAYOUT == 4u ? 0x0C080200u : (LAYOUT == 2u || LAYOUT == 8u) ? 0x0C080400u : 0x0A080200u;   // fo(k), a byte each
        const uint8_t *elfield = reinterpret_cast<const uint8_t *>(table) + 2048u;
        unsigned carry = 0;                                                    // half-tiles begun before the pass
#pragma unroll 1
        for (unsigned e0 = 0; e0 < elements; e0 += 64u) {
            const unsigned e = e0 + lane;
            const unsigned info = e < elements ? elfield[e] : 0u;
            // the half-tile of an element: how many half-tiles have begun up to it (every half-tile has an element)
            const unsigned begun = (unsigned)scan_add((int)(info >> 7)) + carry;
            carry = (unsigned)__builtin_amdgcn_readlane((int)begun, 63);
            if (e < elements) {
                const unsigned h = begun - 1u, q = info & 31u;
                const uint4 ma = *reinterpret_cast<const uint4 *>(&masks[h * 8u]);
                const uint4 mb = *reinterpret_cast<const uint4 *>(&masks[h * 8u + 4u]);
                const unsigned S = ma.x, L = ma.y, X3 = ma.z, D0 = ma.w, D1 = mb.x, Hm = mb.y, Sx1 = mb.z, CS = S & ~L;
                const unsigned under = (1u << q) - 1u;
                const unsigned off = (mb.w & 0xFFFFu) + 4u * popc(L & under) + 2u * popc(L & UL::big32 & under) -
                                     2u * popc(L & UL::small32 & under) + popc(S & under) + popc(CS & under) + popc(X3 & under);
                // bytes from this field to the next start (the mask ends with a set bit)
                const unsigned qn = q + 1u + (unsigned)__builtin_ctz(Sx1 >> q);
                const unsigned len = (qn >> 2) * 16u + ((FO >> (8u * (qn & 3u))) & 0xFFu) - (q >> 2) * 16u - ((FO >> (8u * (q & 3u))) & 0xFFu);
                const bool is_l = ((L >> q) & 1u) != 0u, x3 = ((X3 >> q) & 1u) != 0u;
                // literal run: tag = len - 1 (60 = one length byte follows)
                // copy: copy-2 tag 2 | (len - 1) << 2, offset in the next two bytes; copy-1 (len 4..11, offset < 2048):
                // tag 1 | (len - 4) << 2 | (offset >> 8) << 5, then the offset's low byte
                unsigned dist = 1u + ((D0 >> q) & 1u) + 2u * ((D1 >> q) & 1u);
                if ((Hm >> q) & 1u)
                    dist = (hd_tab[h * 8u + (q >> 2)] >> (8u * (q & 2u))) & 0xFFFFu;        // (index fields are k = 1, 3)
                const unsigned offset = dist * B;
                const unsigned lit = x3 ? 0xF0u | ((len - 1u) << 8) : (len - 1u) << 2;
                const unsigned cpy = x3 ? 2u | ((len - 1u) << 2) | (offset << 8)
                                        : 1u | ((len - 4u) << 2) | ((offset >> 8) << 5) | ((offset & 0xFFu) << 8);
                const unsigned v = is_l ? lit : cpy;
                if (is_l && !x3)
                    put8(out + off, v);
                else
                    put16(out + off, v);
                if (!is_l && x3)
                    put8(out + off + 2, v >> 16);
                const unsigned m = (e * inv) >> 20;
                if (m * G == e)
                    bounds[m] = off | ((h * 128u + (q >> 2) * 16u + ((FO >> (8u * (q & 3u))) & 0xFFu)) << 16);
            }
        }
        __syncthreads();
        if (want_sizes) {
            // fragment table version 4: 24 bits per group -- its compressed bytes | the bytes it produces << 12 -- then
            // the element count
            const unsigned lo = bounds[lane], hi = bounds[lane + 1u];
            const unsigned entry = ((hi & 0xFFFFu) - (lo & 0xFFFFu)) | (((hi >> 16) - (lo >> 16)) << 12);
            const gdst_t at = table_at + lane * 3u;
            put16(at, entry);
            put8(at + 2, entry >> 16);
            if (lane == 0u) {
                put16(table_at + 192, elements);
                put16(table_at + 194, 0u);
            }
        }
    }
}

} // namespace

// layouts: bit 0 = [2,6,4,4] textures present, bit 1 = [4,4], bit 2 = [2,6], bit 3 = [4,4,4,4];
// fused: textures made from the frames' RGBA pictures on the way (HapGpuTexEnc.reserved bits 24..26): bit 0 = scaled
// YCoCg-DXT5, bit 1 = DXT5, bit 2 = DXT1, bit 3 = RGTC1
extern "C" int hapgpu_launch_snappy_compress_blocks(const HapGpuFrameEnc *frames, unsigned frame_count,
                                                    unsigned max_frags_per_texture, unsigned textures, void *slots,
                                                    unsigned slot_stride, uint32_t *frag_sizes, uint8_t *group_tables,
                                                    unsigned layouts, unsigned fused, hipStream_t stream)
{
    if (frame_count == 0 || max_frags_per_texture == 0 || (layouts | fused) == 0)
        return 0;
    const dim3 grid(max_frags_per_texture, textures, frame_count), block(64);
#define HAP_LAUNCH_BLOCKS(L, F)                                                                                                  \
    hipLaunchKernelGGL((snappy_compress_blocks_kernel<L, F>), grid, block, 0, stream, frames, (uint8_t *)slots, slot_stride,    \
                       frag_sizes, group_tables)
    if (fused & 1u)
        HAP_LAUNCH_BLOCKS(4u, hapbc::kFmtYCoCg);
#ifndef SCB_ONLY_FUSED_YCOCG          // (instruction-count studies, tools/isa_by_line.py: that instantiation alone)
    if (fused & 2u)
        HAP_LAUNCH_BLOCKS(4u, hapbc::kFmtDXT5);
    if (fused & 12u)          // (DXT1 / RGTC1: the fused form was no faster than the two passes, hap_batch.c; not instantiated)
        return 1;
    if (layouts & 1u)
        HAP_LAUNCH_BLOCKS(4u, -1);
    if (layouts & 2u)
        HAP_LAUNCH_BLOCKS(2u, -1);
    if (layouts & 4u)
        HAP_LAUNCH_BLOCKS(6u, -1);
    if (layouts & 8u)
        HAP_LAUNCH_BLOCKS(8u, -1);
#endif
#undef HAP_LAUNCH_BLOCKS
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
