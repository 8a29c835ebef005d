// snappy_compress.hip -- per-fragment Snappy compressors for gfx950.
//
// Replaces the snappy_compress call-out of the reference's chunk loop (hap.c:448-476, call at
// hap.c:453).  libsnappy itself compresses independent 64 KiB fragments with a per-fragment
// hash table; here a fragment is 2^frag_log2 bytes (default 8 KiB) so that
// fragment + hash table fit in LDS many times per CU.  Output is ordinary Snappy elements
// (literal / copy-1 / copy-2); a chunk's stream is varint(chunk bytes) followed by its fragments'
// element runs, concatenated by the pack/gather kernels (frame_pack.hip).  The produced bytes
// differ from libsnappy's (Snappy encoding is not unique); parity is defined as: the reference
// decoder reproduces the input exactly.
//
// Block textures (DXT1 / DXT5 / YCoCg-DXT5, large RGTC1 planes) go to the block-per-lane kernels of
// snappy_compress_blocks.hip; this file holds the compressor for everything else:
//   snappy_compress_wg_kernel     one 256-thread workgroup per fragment (fragment + hash table in LDS, synchronous
//                                 rounds), a lane owns a 1 / 2 / 4-byte position.
// Per tile: (a) candidates, all lanes at once: the most recent earlier position with the same hash
// (LDS table, updated between rounds), and fixed block-pitch distances through wave ballots (equality bit per
// lane, run length = count-trailing-ones of the shifted mask); (b) greedy left-to-right selection on the
// scalar unit; (c) emission, all lanes at once: every lane knows the bytes it emits, offsets come from a DPP
// prefix sum, each lane stores its own element bytes at their final place in the fragment's slot.
// HBM traffic: fragment read once, compressed bytes written once.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include "hapgpu_abi.h"
#include "measurement_guard.h"

namespace {



// 4 bytes at an arbitrary LDS byte offset (two aligned dword reads + byte align)
__device__ __forceinline__ unsigned lds_load32(const uint32_t *words, unsigned byte_off)
{
    const unsigned w = byte_off >> 2;
    return __builtin_amdgcn_alignbyte(words[w + 1], words[w], byte_off & 3u);
}

// wave-wide predicate mask; keep the argument a single compare so that it lowers to one v_cmp
__device__ __forceinline__ unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global
// store (s_waitcnt vmcnt(0)); the round barriers of the workgroup kernel exchange LDS data only, and waiting
// for the element bytes to reach L2 twice per round was the largest single cost of the kernel.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ unsigned bits_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// run of consecutive 1 bits starting at bit `lane` of the 128-bit value next:cur, capped at 64
__device__ __forceinline__ unsigned run_from(unsigned long long cur, unsigned long long next, unsigned lane)
{
    const unsigned long long a = ~(cur >> lane);
    const unsigned avail = 64u - lane;
    unsigned r = a ? (unsigned)__builtin_ctzll(a) : 64u;
    if (r >= avail) {
        const unsigned long long b = ~next;
        r = avail + (b ? (unsigned)__builtin_ctzll(b) : 64u);
    }
    return min(r, 64u);
}


// ------------------------------------------------------------------------------------------
// workgroup-per-fragment compressor
// ------------------------------------------------------------------------------------------
//
// Same element stream rules as the kernel above, restructured for throughput:
//   * 4 wavefronts share one fragment (data + hash table in LDS once, 4x the waves per CU);
//     in every ROUND wave w takes the 128-byte supertile 4*round + w (two 64-byte tiles).
//   * rounds are synchronous: all lookups of a round read the hash table as it was after the
//     previous round, then all waves insert with LDS atomicMax (u32 entries, the most recent
//     position wins) -- the output does not depend on wave timing (deterministic).
//   * copies never cross a supertile boundary, so supertiles are independent; their sizes are
//     exchanged through LDS at the round barrier and every wave writes its bytes straight to
//     their final position (no staging, no compaction pass).
//   * lanes hold 1, 2 or 4 bytes (GRAN): with 16/32-bit granular positions, offsets and lengths a wave covers
//     128 / 256 bytes per tile, and the decoder moves as much per lane (the fragment table records it);
//   * candidates: the hash probe of both tiles is verified side by side, 16 bytes per compare and without a
//     branch (the two LDS round trips overlap), plus four block-pitch distances through ballots;
//   * what depends only on the tile (range masks, kinds of elements, byte counts) is scalar work; lanes get
//     the masks back through inverse ballots; the greedy selection is a hand-written scalar loop;
//   * every lane's element leaves as one value with 16-bit stores; the round barriers order LDS traffic only.


#ifndef HAP_WG_WAVES
#define HAP_WG_WAVES 4
#endif
constexpr unsigned kWgWaves = HAP_WG_WAVES;
#ifndef HAP_WG_SUBS
#define HAP_WG_SUBS 2
#endif
constexpr unsigned kSubs = HAP_WG_SUBS;   // tiles per wave per round (one "supertile"; copies never cross it)
constexpr int kFixed = 4;      // candidates at 1..4 block pitches (8-byte or 16-byte blocks)
#ifndef HAP_WG_HASH_BITS
#define HAP_WG_HASH_BITS 11
#endif
constexpr unsigned kWgHashBits = HAP_WG_HASH_BITS;
constexpr unsigned kWgHashEntries = 1u << kWgHashBits;

__device__ __forceinline__ int cdpp_row_shr(int v, int n)
{
    switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    }
}

__device__ __forceinline__ int cwave_scan_max(int v)   // inclusive, values >= 0
{
    v = max(v, cdpp_row_shr(v, 1));
    v = max(v, cdpp_row_shr(v, 2));
    v = max(v, cdpp_row_shr(v, 4));
    v = max(v, cdpp_row_shr(v, 8));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));
    return v;
}

// does any lane of any tile still extend its hash match?  (scalar answer)
__device__ __forceinline__ bool any_more(const bool (&more)[kSubs])
{
    unsigned long long m = 0;
#pragma unroll
    for (unsigned i = 0; i < kSubs; i++)
        m |= __builtin_amdgcn_ballot_w64(more[i]);
    return m != 0;
}

// Greedy left-to-right choice of non-overlapping copies on the scalar unit: from `cursor`, take the first
// candidate lane s at or after it (bit s of `cand`), mark it in `sel`, continue at next_free[s] (the lane after
// that copy's last position), until lane 64 is passed or no candidate is left.  Hand-written because the
// compiler's structured form of this loop costs twice the scalar instructions, and this loop is where the
// kernel's scalar time goes:
//     while (cursor < 64) { rest = cand >> cursor; if (!rest) break;
//                           s = cursor + ctz(rest); sel |= 1 << s; cursor = readlane(next_free, s); }
__device__ __forceinline__ void greedy_select(unsigned long long cand, unsigned next_free, unsigned &cursor,
                                              unsigned long long &sel)
{
    unsigned long long rest;
    unsigned step;
    asm volatile("s_cmp_lt_u32 %[cur], 64\n\t"
                 "s_cbranch_scc0 2f\n"
                 "1:\n\t"
                 "s_lshr_b64 %[rest], %[cand], %[cur]\n\t"      // SCC = (rest != 0)
                 "s_cbranch_scc0 2f\n\t"
                 "s_ff1_i32_b64 %[step], %[rest]\n\t"
                 "s_add_u32 %[cur], %[cur], %[step]\n\t"
                 "s_bitset1_b64 %[sel], %[cur]\n\t"
                 "v_readlane_b32 %[cur], %[next], %[cur]\n\t"
                 "s_cmp_lt_u32 %[cur], 64\n\t"
                 "s_cbranch_scc1 1b\n"
                 "2:"
                 : [cur] "+s"(cursor), [sel] "+s"(sel), [rest] "=&s"(rest), [step] "=&s"(step)
                 : [cand] "s"(cand), [next] "v"(next_free)
                 : "scc");
}

// number of equal leading bytes (0..16) of the 16 bytes at LDS byte offsets a and b
__device__ __forceinline__ unsigned match16(const uint32_t *dw, unsigned a, unsigned b)
{
    const unsigned wa = a >> 2, sa = a & 3u, wb = b >> 2, sb = b & 3u;
    const unsigned a0 = dw[wa], a1 = dw[wa + 1], a2 = dw[wa + 2], a3 = dw[wa + 3], a4 = dw[wa + 4];
    const unsigned b0 = dw[wb], b1 = dw[wb + 1], b2 = dw[wb + 2], b3 = dw[wb + 3], b4 = dw[wb + 4];
    const unsigned d0 = __builtin_amdgcn_alignbyte(a1, a0, sa) ^ __builtin_amdgcn_alignbyte(b1, b0, sb);
    const unsigned d1 = __builtin_amdgcn_alignbyte(a2, a1, sa) ^ __builtin_amdgcn_alignbyte(b2, b1, sb);
    const unsigned d2 = __builtin_amdgcn_alignbyte(a3, a2, sa) ^ __builtin_amdgcn_alignbyte(b3, b2, sb);
    const unsigned d3 = __builtin_amdgcn_alignbyte(a4, a3, sa) ^ __builtin_amdgcn_alignbyte(b4, b3, sb);
    return d0 ? ((unsigned)__builtin_ctz(d0) >> 3)
         : d1 ? 4u + ((unsigned)__builtin_ctz(d1) >> 3)
         : d2 ? 8u + ((unsigned)__builtin_ctz(d2) >> 3)
         : d3 ? 12u + ((unsigned)__builtin_ctz(d3) >> 3)
              : 16u;
}

// two bytes at any byte address (global memory takes unaligned accesses)
struct __attribute__((packed)) packed_u16 { uint16_t v; };
__device__ __forceinline__ void store16(uint8_t *p, unsigned v) { reinterpret_cast<packed_u16 *>(p)->v = (uint16_t)v; }

// GRAN: bytes per lane: 1, 2 or 4 (positions, offsets and lengths all multiples of GRAN).
// FL: log2 of the fragment size when known at compile time (the default, 13), 0 = run-time value: with a fixed
//     size the LDS is a static array and every address offset a compile-time constant.
template <unsigned GRAN, unsigned FL>
__global__ __launch_bounds__(64 * kWgWaves) void snappy_compress_wg_kernel(const HapGpuFrameEnc *__restrict__ frames,
                                                                unsigned frag_log2_arg, uint8_t *__restrict__ slots,
                                                                unsigned slot_stride, uint32_t *__restrict__ frag_sizes)
{
    constexpr unsigned kStaticBytes = FL ? (1u << FL) + 32u + kWgHashEntries * 4u + kWgWaves * 4u : 16u;
    extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
    __shared__ __attribute__((aligned(16))) uint8_t static_lds[kStaticBytes];
    uint8_t *const smem = FL ? static_lds : dynamic_lds;
    const unsigned frag_log2 = FL ? FL : frag_log2_arg;
    const unsigned frag_bytes = 1u << frag_log2;
    uint32_t *dataw = reinterpret_cast<uint32_t *>(smem);                        // frag_bytes + 32
    const uint8_t *data = smem;
    uint32_t *table = reinterpret_cast<uint32_t *>(smem + frag_bytes + 32);      // kWgHashEntries
    uint32_t *roundsz = table + kWgHashEntries;                                  // kWgWaves

    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));     // uniform: scalar register
    // one trip to memory for the whole descriptor (field-by-field reads with the early exits between them
    // cost a scalar-load round trip each)
    const HapGpuTexEnc tex = frames[blockIdx.z].tex[blockIdx.y < 2u ? blockIdx.y : 0u];
    const unsigned tex_count = frames[blockIdx.z].tex_count;
    const unsigned x = blockIdx.x;
    // (one combined test, no short-circuit: every field is requested before the first wait)
    if ((blockIdx.y >= tex_count) | (tex.compressor != 1u) | ((1u << (tex.reserved & 0xFFu)) != GRAN) |
        (((tex.reserved >> 16) & 0xFu) != 0u) |        // field-per-lane textures belong to the kernel below
        (x >= tex.chunk_count * tex.frags_per_chunk) | (tex.src == 0) | (tex.chunk_bytes == 0))
        return;
    const unsigned chunk = x / tex.frags_per_chunk, j = x - chunk * tex.frags_per_chunk;
    const unsigned begin = j << frag_log2;
    const unsigned n = min(frag_bytes, tex.chunk_bytes - begin);
    const uint8_t *src = (const uint8_t *)tex.src + (size_t)chunk * tex.chunk_bytes + begin;
    const unsigned f = tex.frag_first + x;
    const unsigned window = (tex.reserved >> 8) ? (tex.reserved >> 8) * 256u : 0xFFFFFFFFu;
    uint8_t *out = slots + (size_t)f * slot_stride;

    if (((uintptr_t)src & 15u) == 0) {
        // all loads of a batch are in flight before the first one is waited for
        constexpr unsigned kBatch = 4;
        constexpr unsigned kStride = 1024u * kWgWaves;
        for (unsigned i0 = tid * 16u; i0 < n + 32u; i0 += kBatch * kStride) {
            uint4 v[kBatch];
#pragma unroll
            for (unsigned u = 0; u < kBatch; u++) {
                const unsigned i = i0 + u * kStride;
                v[u] = make_uint4(0, 0, 0, 0);
                if (i + 16u <= n)
                    v[u] = *reinterpret_cast<const uint4 *>(src + i);
            }
#pragma unroll
            for (unsigned u = 0; u < kBatch; u++) {
                const unsigned i = i0 + u * kStride;
                if (i < n && i + 16u > n) {                 // ragged end (never for whole blocks)
                    unsigned w[4] = {0, 0, 0, 0};
                    for (unsigned k = 0; i + k < n; k++)
                        w[k >> 2] |= (unsigned)src[i + k] << (8 * (k & 3));
                    v[u] = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (i < n + 32u)
                    *reinterpret_cast<uint4 *>(smem + i) = v[u];
            }
        }
    } else {
        for (unsigned i = tid; i < n + 32u; i += 64u * kWgWaves)
            smem[i] = i < n ? src[i] : (uint8_t)0;
    }
    for (unsigned i = tid; i < kWgHashEntries; i += 64u * kWgWaves)
        table[i] = 0u;
    __syncthreads();
    using val_t = typename std::conditional<GRAN == 4, unsigned long long, unsigned>::type;
    constexpr unsigned TB = 64u * GRAN;                 // bytes per tile
    constexpr unsigned GL = GRAN == 4 ? 2u : GRAN == 2 ? 1u : 0u;     // log2(GRAN)
    const uint16_t *data16 = reinterpret_cast<const uint16_t *>(smem);
    const uint32_t *data32 = reinterpret_cast<const uint32_t *>(smem);
    const unsigned tiles = (n + TB - 1u) / TB, supers = (tiles + kSubs - 1u) / kSubs;
    unsigned round_base = 0;
    // DXT1 / RGTC1 textures are arrays of 8-byte blocks, everything else 16-byte blocks (hap.c:287-294)
    const unsigned pitch_log2 = (tex.format_nibble == 0xBu || tex.format_nibble == 0x1u) ? 3u : 4u;
    const bool upper = lane >= 32u;
    const unsigned lane31 = lane & 31u;
    (void)data16;
    (void)data32;

    // Everything that depends only on the tile (k, tile_base, range masks, element kind masks) lives on the
    // scalar unit: `wave` is uniform, predicates reach the lanes through inverse ballots.
    for (unsigned base = 0; base < supers; base += kWgWaves) {
        const unsigned k = base + wave;
        const bool have = k < supers;
        // per-tile emission plan, kept in registers across the round barrier
        // (a lane's element is at most GRAN + 2 bytes: they travel as one value, low byte first)
        unsigned p_hash[kSubs] = {}, p_at[kSubs] = {};
        val_t p_val[kSubs] = {};
        unsigned long long m_e0[kSubs] = {}, m_e1[kSubs] = {}, m_e2[kSubs] = {}, m_insert[kSubs] = {};
        unsigned total = 0;
        if (have) {
            const unsigned super_end = min(n, (kSubs * k + kSubs) * TB);
            // equality ballots for the fixed distances (block pitches of DXT data)
            unsigned long long eq[kFixed][kSubs], in_mask[kSubs];
#pragma unroll
            for (int sub = 0; sub < (int)kSubs; sub++) {
                const unsigned tile_base = (kSubs * k + sub) * TB;
                const unsigned p = tile_base + GRAN * lane;
                const unsigned cnt = tile_base < n ? min(64u, (n - tile_base) >> GL) : 0u;      // lanes with p < n
                in_mask[sub] = cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull);
                const unsigned here = GRAN == 4 ? data32[p >> 2] : GRAN == 2 ? (unsigned)data16[p >> 1] : (unsigned)data[p];
#pragma unroll
                for (int d = 0; d < kFixed; d++) {
                    const unsigned dist = (unsigned)(d + 1) << pitch_log2;
                    const unsigned back = p >= dist ? p - dist : 0u;
                    const unsigned there = GRAN == 4 ? data32[back >> 2] : GRAN == 2 ? (unsigned)data16[back >> 1] : (unsigned)data[back];
                    unsigned long long reachable = ~0ull;                                        // lanes with p >= dist
                    if (tile_base < dist) {
                        const unsigned first = (dist - tile_base) >> GL;
                        reachable = first >= 64u ? 0ull : ~0ull << first;
                    }
                    eq[d][sub] = ballot64(here == there) & in_mask[sub] & reachable;
                }
            }
            // ---- candidates, both tiles side by side so that their LDS round trips overlap ----
            unsigned room2[kSubs], cur2[kSubs], h2[kSubs], hlen[kSubs], hoff[kSubs], cand2[kSubs];
            unsigned long long mask4[kSubs];
            bool more[kSubs];
#pragma unroll
            for (int sub = 0; sub < (int)kSubs; sub++) {
                const unsigned tile_base = (kSubs * k + sub) * TB;
                const unsigned p = tile_base + GRAN * lane;
                // bytes a copy starting here may span: up to the supertile end, 0 beyond the data
                room2[sub] = min(64u, super_end > p ? super_end - p : 0u);
                cur2[sub] = lds_load32(dataw, p);
                h2[sub] = (cur2[sub] * 0x1e35a7bdu) >> (32u - kWgHashBits);
                const unsigned cnt4 = n >= tile_base + 4u ? min(64u, ((n - 4u - tile_base) >> GL) + 1u) : 0u;   // lanes with p + 4 <= n
                mask4[sub] = cnt4 >= 64u ? ~0ull : ((1ull << cnt4) - 1ull);
            }
#pragma unroll
            for (int sub = 0; sub < (int)kSubs; sub++) {
                const unsigned p = (kSubs * k + sub) * TB + GRAN * lane;
                hlen[sub] = 0;
                hoff[sub] = 0;
                more[sub] = false;
                cand2[sub] = 0;
                // hash candidate: the first 16 bytes are compared at once, without branching
                const unsigned c = table[h2[sub]];
                // (a match window, when the texture asks for one, keeps hash candidates close: the decoder then
                // needs only half a fragment of LDS)
                const bool valid = __builtin_amdgcn_inverse_ballot_w64(mask4[sub]) && c < p && room2[sub] >= 4u &&
                                   p - c <= window;
                cand2[sub] = valid ? c : 0u;
                const unsigned m = match16(dataw, cand2[sub], p);
                if (valid && m >= 4u) {
                    hlen[sub] = min(m, room2[sub]);
                    hoff[sub] = p - c;
                    more[sub] = m == 16u && room2[sub] > 16u;
                }
            }
            // longer hash matches (uncommon): both tiles advance together, 16 bytes per step
            while (any_more(more)) {
#pragma unroll
                for (int sub = 0; sub < (int)kSubs; sub++) {
                    if (more[sub]) {
                        const unsigned p = (kSubs * k + sub) * TB + GRAN * lane;
                        const unsigned m = match16(dataw, cand2[sub] + hlen[sub], p + hlen[sub]);
                        hlen[sub] = min(hlen[sub] + m, room2[sub]);
                        more[sub] = m == 16u && hlen[sub] < room2[sub];
                    }
                }
            }
            unsigned best_len2[kSubs], best_off2[kSubs];
#pragma unroll
            for (int sub = 0; sub < (int)kSubs; sub++) {
                const unsigned room = room2[sub];
                // best candidate as one key: (length << 3) | priority, nearer fixed distances win ties,
                // the hash candidate (priority 0) only when strictly longer
                unsigned best_key = (hlen[sub] & ~(GRAN - 1u)) << 3;
#pragma unroll
                for (int d = kFixed - 1; d >= 0; d--) {
                    const unsigned long long c = eq[d][sub], nx = sub + 1 < (int)kSubs ? eq[d][sub + 1 < (int)kSubs ? sub + 1 : sub] : 0ull;
                    if (c == 0ull)                    // (uniform) nothing of this tile repeats at this distance
                        continue;
                    unsigned l;
                    if (GRAN >= 2) {
                        // run of set bits starting at this lane, capped at 32 lanes (>= 64 bytes): one funnel shift
                        const unsigned lo = upper ? (unsigned)(c >> 32) : (unsigned)c;
                        const unsigned hi = upper ? (unsigned)nx : (unsigned)(c >> 32);
                        const unsigned inv = ~__builtin_amdgcn_alignbit(hi, lo, lane31);
                        l = min(inv ? (unsigned)__builtin_ctz(inv) : 32u, room >> GL);
                    } else {
                        l = min(run_from(c, nx, lane), room);
                    }
                    best_key = max(best_key, (l << (3u + GL)) | (unsigned)(d + 1));
                }
                const unsigned prio = best_key & 7u;
                best_len2[sub] = best_key >> 3;
                best_off2[sub] = prio ? prio << pitch_log2 : hoff[sub];
            }
            unsigned skip = 0;
#pragma unroll
            for (int sub = 0; sub < (int)kSubs; sub++) {
                const unsigned best_len = best_len2[sub], best_off = best_off2[sub];
                const unsigned cur = cur2[sub], h = h2[sub];
                // greedy selection: the scalar unit hops from chosen copy to chosen copy
                const unsigned long long cand_mask = ballot64(best_len >= 4u);
                unsigned long long sel = 0;
                unsigned cursor = (unsigned)__builtin_amdgcn_readfirstlane((int)skip);   // first position not yet covered
                const unsigned next_free = lane + (best_len >> GL);     // ... after taking this lane's copy
                greedy_select(cand_mask, next_free, cursor, sel);
                const unsigned carry = cursor > 64u ? cursor - 64u : 0u;
                // covered <=> some chosen copy (or the carry-in) spans the position
                const int reach = cwave_scan_max(__builtin_amdgcn_inverse_ballot_w64(sel) ? (int)next_free : 0);
                const unsigned long long skipmask = skip >= 64u ? ~0ull : ((1ull << skip) - 1ull);
                const unsigned long long lit = ~(ballot64((unsigned)reach > lane) | skipmask) & in_mask[sub];
                skip = carry;
                const unsigned long long starts = lit & ~(lit << 1);
                unsigned run = 0;                          // literal run length in BYTES
                if (__builtin_amdgcn_inverse_ballot_w64(starts)) {
                    const unsigned long long a = ~(lit >> lane);
                    run = GRAN * (a ? (unsigned)__builtin_ctzll(a) : 64u);
                }
                const unsigned long long longrun = ballot64(run > 60u);                 // subset of starts
                const unsigned long long copy1 = ballot64(best_len < 12u) & ballot64(best_off < 2048u);
                const unsigned long long copy2 = sel & ~copy1;
                // bytes a lane emits: literal GRAN (+1 at a run start, +2 when the run is long), copy 2 or 3;
                // as bit planes of that count, composed on the scalar unit
                unsigned long long e0, e1, e2;
                if (GRAN == 1) {
                    e0 = (lit & ~starts) | longrun | copy2; e1 = starts | sel; e2 = 0ull;
                } else if (GRAN == 2) {
                    e0 = (starts & ~longrun) | copy2; e1 = (lit & ~longrun) | sel; e2 = longrun;
                } else {
                    e0 = (starts & ~longrun) | copy2; e1 = longrun | sel; e2 = lit;
                }
                p_at[sub] = total + bits_below(e0) + 2u * bits_below(e1) + (GRAN >= 2 ? 4u * bits_below(e2) : 0u);
                total += (unsigned)__popcll(e0) + 2u * (unsigned)__popcll(e1) + 4u * (unsigned)__popcll(e2);
                // the element bytes of this lane (whatever lies beyond its byte count is never stored)
                val_t v = (val_t)cur;                                                   // literal inside a run
                if (__builtin_amdgcn_inverse_ballot_w64(starts))
                    v = run > 60u ? (val_t)(0xF0u | ((run - 1u) << 8)) | ((val_t)cur << 16)
                                  : (val_t)((run - 1u) << 2) | ((val_t)cur << 8);
                if (__builtin_amdgcn_inverse_ballot_w64(sel))
                    v = (val_t)(__builtin_amdgcn_inverse_ballot_w64(copy1)
                                    ? (1u | ((best_len - 4u) << 2) | ((best_off >> 8) << 5) | ((best_off & 0xFFu) << 8))
                                    : (2u | ((best_len - 1u) << 2) | (best_off << 8)));
                p_val[sub] = v;
                p_hash[sub] = h;
                m_e0[sub] = e0;
                m_e1[sub] = e1;
                m_e2[sub] = e2;
                m_insert[sub] = (lit | sel) & mask4[sub];       // only element starts are remembered (see the single-wave kernel)
            }
        }
        if (lane == 0)
            roundsz[wave] = total;
        lds_barrier();
        unsigned my_base = round_base, all = 0;
#pragma unroll
        for (unsigned w = 0; w < kWgWaves; w++) {
            const unsigned sz = roundsz[w];
            if (w < wave)
                my_base += sz;
            all += sz;
        }
        round_base += all;
        if (have) {
#pragma unroll
            for (int sub = 0; sub < (int)kSubs; sub++) {
                const unsigned p = (kSubs * k + sub) * TB + GRAN * lane;
                // byte count of the lane's element = e0 + 2 e1 + 4 e2: whole 16-bit pieces first, then the odd byte
                uint8_t *dst = out + my_base + p_at[sub];
                const val_t v = p_val[sub];
                {
                const unsigned long long e0 = m_e0[sub], e1 = m_e1[sub], e2 = m_e2[sub];
                if (__builtin_amdgcn_inverse_ballot_w64(e1 | e2))
                    store16(dst, (unsigned)v);
                if (GRAN >= 2 && __builtin_amdgcn_inverse_ballot_w64(e2))
                    store16(dst + 2, (unsigned)(v >> 16));
                if (GRAN == 4 && __builtin_amdgcn_inverse_ballot_w64(e2 & e1))
                    store16(dst + 4, (unsigned)((unsigned long long)v >> 32));
                if (GRAN == 1 && __builtin_amdgcn_inverse_ballot_w64(e0 & ~e1))
                    dst[0] = (uint8_t)v;
                if (__builtin_amdgcn_inverse_ballot_w64(e0 & e1 & ~e2))
                    dst[2] = (uint8_t)(v >> 16);
                if (GRAN == 4 && __builtin_amdgcn_inverse_ballot_w64(e0 & e2 & ~e1))
                    dst[4] = (uint8_t)((unsigned long long)v >> 32);
                }
                if (__builtin_amdgcn_inverse_ballot_w64(m_insert[sub]))
                    atomicMax(&table[p_hash[sub]], p);
            }
        }
        lds_barrier();
    }
    if (tid == 0)
        frag_sizes[f] = round_base;
}


} // namespace

extern "C" int hapgpu_launch_snappy_compress_blocks(const HapGpuFrameEnc *frames, unsigned frame_count,
                                                    unsigned max_frags_per_texture, unsigned textures, void *slots,
                                                    unsigned slot_stride, uint32_t *frag_sizes, uint8_t *group_tables,
                                                    unsigned layouts, unsigned fused, hipStream_t stream);

extern "C" int hapgpu_launch_snappy_compress(const HapGpuFrameEnc *frames, unsigned frame_count,
                                             unsigned max_frags_per_texture, unsigned frag_log2, void *slots,
                                             unsigned slot_stride, uint32_t *frag_sizes, uint8_t *group_tables,
                                             unsigned granularity_mask, hipStream_t stream)
{
    if (frame_count == 0 || max_frags_per_texture == 0)
        return 0;
    if (frag_log2 < 10 || frag_log2 > 16)
        return 1;
    {
        const unsigned lds2 = (1u << frag_log2) + 32u + kWgHashEntries * 4u + kWgWaves * 4u;
        if (lds2 > 65536u) {
            static bool once2 = false;
            if (!once2) {
                if (hipFuncSetAttribute((const void *)snappy_compress_wg_kernel<1u, 0u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
                    hipFuncSetAttribute((const void *)snappy_compress_wg_kernel<2u, 0u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
                    hipFuncSetAttribute((const void *)snappy_compress_wg_kernel<4u, 0u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
                    return 4;
                once2 = true;
            }
        }
        const unsigned textures = ((granularity_mask >> 8) & 0xFFu) == 1u ? 1u : 2u;      // bits 8..15: textures per frame (0 = unknown)
        const unsigned fused = (granularity_mask >> 16) & 0xFu;                 // bits 16..19: textures made from RGBA on the way
        const dim3 grid(max_frags_per_texture, textures, frame_count), block(64 * kWgWaves);
#define HAP_LAUNCH_COMPRESS(G)                                                                                                  \
        do {                                                                                                                    \
            if (frag_log2 == 13u)                                                                                               \
                hipLaunchKernelGGL((snappy_compress_wg_kernel<G, 13u>), grid, block, 0, stream, frames, frag_log2,              \
                                   (uint8_t *)slots, slot_stride, frag_sizes);                                                  \
            else                                                                                                                \
                hipLaunchKernelGGL((snappy_compress_wg_kernel<G, 0u>), grid, block, lds2, stream, frames, frag_log2,            \
                                   (uint8_t *)slots, slot_stride, frag_sizes);                                                  \
        } while (0)
        // block textures: the block-per-lane kernels of snappy_compress_blocks.hip
        if ((granularity_mask & 0xF0u) | fused) {
            if (frag_log2 != 13u)
                return 1;
            const unsigned layouts = ((granularity_mask & 32u) ? 1u : 0u) | ((granularity_mask & 64u) ? 2u : 0u) |
                                     ((granularity_mask & 16u) ? 4u : 0u) | ((granularity_mask & 128u) ? 8u : 0u);
            if (hapgpu_launch_snappy_compress_blocks(frames, frame_count, max_frags_per_texture, textures, slots, slot_stride,
                                                     frag_sizes, group_tables, layouts, fused, stream))
                return 4;
        }
        if (granularity_mask & 1u)
            HAP_LAUNCH_COMPRESS(1u);
        if (granularity_mask & 2u)
            HAP_LAUNCH_COMPRESS(2u);
        if (granularity_mask & 4u)
            HAP_LAUNCH_COMPRESS(4u);
#undef HAP_LAUNCH_COMPRESS
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
}
