// snappy_compress.hip -- per-fragment Snappy compressor for gfx950.
//
// Replaces the snappy_compress call-out of the reference's chunk loop (hap.c:448-476, call at
// hap.c:453).  libsnappy itself compresses independent 64 KiB fragments with a per-fragment
// hash table; here a fragment is 2^frag_log2 bytes (default 16 KiB) so that
// fragment + hash table fit in LDS several times per CU, and one wavefront compresses one
// fragment.  Output is ordinary Snappy elements (literal / copy-1 / copy-2); a chunk's stream is
// varint(chunk bytes) followed by its fragments' element runs, concatenated by the pack/gather
// kernels (frame_pack.hip).  The produced bytes differ from libsnappy's (Snappy encoding is not
// unique); parity is defined as: the reference decoder reproduces the input exactly.
//
// Per 64-byte tile, lane l owns input position p = tile*64 + l:
//   1. match finding, all lanes at once: (a) hash of the 4 bytes at p -> most recent earlier
//      position with that hash (LDS u16 table, updated after the lookup), verified and extended
//      up to 64 bytes; (b) fixed distances 8 and 16 (the block pitch of DXT data) through wave
//      ballots: equality bit per position, run length = count-trailing-ones of the shifted mask.
//   2. greedy selection: the scalar unit walks the ballot of "match >= 4" left to right,
//      skipping the bytes each chosen copy covers (v_readlane for the length).
//   3. emission, all lanes at once: uncovered positions are literal bytes; every lane knows the
//      number of bytes it emits (0..3), offsets come from two ballots + mbcnt, and each lane
//      stores its own tag/data bytes.
// HBM traffic: fragment read once, compressed bytes written once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "hapgpu_abi.h"

namespace {

#ifndef HAP_HASH_BITS
#define HAP_HASH_BITS 12
#endif
constexpr unsigned kHashBits = HAP_HASH_BITS;
constexpr unsigned kHashEntries = 1u << kHashBits;

__device__ __forceinline__ unsigned uniform(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    return ((unsigned long long)uniform((unsigned)(v >> 32)) << 32) | uniform((unsigned)v);
}

// 4 bytes at an arbitrary LDS byte offset (two aligned dword reads + byte align)
__device__ __forceinline__ unsigned lds_load32(const uint32_t *words, unsigned byte_off)
{
    const unsigned w = byte_off >> 2;
    return __builtin_amdgcn_alignbyte(words[w + 1], words[w], byte_off & 3u);
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

__global__ __launch_bounds__(64) void snappy_compress_kernel(const HapGpuFrameEnc *__restrict__ frames,
                                                             unsigned frag_log2, uint8_t *__restrict__ slots,
                                                             unsigned slot_stride, uint32_t *__restrict__ frag_sizes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const unsigned frag_bytes = 1u << frag_log2;
    uint32_t *dataw = reinterpret_cast<uint32_t *>(smem);                 // frag_bytes + 16
    const uint8_t *data = smem;
    uint16_t *table = reinterpret_cast<uint16_t *>(smem + frag_bytes + 16);

    const unsigned lane = threadIdx.x;
    const HapGpuFrameEnc &frame = frames[blockIdx.z];
    if (blockIdx.y >= frame.tex_count)
        return;
    const HapGpuTexEnc &tex = frame.tex[blockIdx.y];
    if (tex.compressor != 1u)
        return;
    const unsigned x = blockIdx.x;
    if (x >= tex.chunk_count * tex.frags_per_chunk)
        return;
    const unsigned chunk = x / tex.frags_per_chunk, j = x - chunk * tex.frags_per_chunk;
    const unsigned begin = j << frag_log2;
    const unsigned n = min(frag_bytes, tex.chunk_bytes - begin);
    const uint8_t *src = (const uint8_t *)tex.src + (size_t)chunk * tex.chunk_bytes + begin;
    const unsigned f = tex.frag_first + x;
    uint8_t *out = slots + (size_t)f * slot_stride;

    // ---- stage the fragment and clear the hash table ----
    if (((uintptr_t)src & 15u) == 0) {
        for (unsigned i = lane * 16u; i < n + 16u; i += 1024u) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i + 16u <= n) {
                v = *reinterpret_cast<const uint4 *>(src + i);
            } else if (i < n) {
                unsigned w[4] = {0, 0, 0, 0};
                for (unsigned k = 0; i + k < n; k++)
                    w[k >> 2] |= (unsigned)src[i + k] << (8 * (k & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4 *>(smem + i) = v;
        }
    } else {
        for (unsigned i = lane; i < n + 16u; i += 64u)
            smem[i] = i < n ? src[i] : (uint8_t)0;
    }
    for (unsigned i = lane; i < kHashEntries / 2; i += 64u)
        reinterpret_cast<uint32_t *>(table)[i] = 0u;
    __syncthreads();

    const unsigned tiles = (n + 63u) / 64u;
    unsigned out_pos = 0;       // bytes emitted so far
    unsigned skip = 0;          // leading positions of the current tile covered by an earlier copy

    // equality ballots for the fixed distances, one tile ahead
    auto eq_mask = [&](unsigned tile, unsigned d) -> unsigned long long {
        const unsigned p = tile * 64u + lane;
        const bool e = tile < tiles && p >= d && p < n && data[p] == data[p - d];
        return __ballot(e);
    };
    unsigned long long m8 = eq_mask(0, 8), m16 = eq_mask(0, 16);

    for (unsigned t = 0; t < tiles; t++) {
        const unsigned p = t * 64u + lane;
        const bool in_range = p < n;
        const unsigned room = in_range ? min(64u, n - p) : 0u;      // longest match allowed here
        const unsigned long long n8 = eq_mask(t + 1, 8), n16 = eq_mask(t + 1, 16);

        // ---- (a) hash candidate ----
        unsigned best_len = 0, best_off = 0, my_hash = 0xFFFFFFFFu;
        if (p + 4u <= n) {
            const unsigned cur = lds_load32(dataw, p);
            const unsigned h = (cur * 0x1e35a7bdu) >> (32u - kHashBits);
            const unsigned cand = table[h];
            my_hash = h;
            if (cand < p && lds_load32(dataw, cand) == cur) {
                unsigned l = 4;
                while (l < room) {
                    const unsigned diff = lds_load32(dataw, cand + l) ^ lds_load32(dataw, p + l);
                    if (diff) {
                        l += (unsigned)__builtin_ctz(diff) >> 3;
                        break;
                    }
                    l += 4;
                }
                best_len = min(l, room);
                best_off = p - cand;
            }
        }
        // ---- (b) fixed distances ----
        {
            const unsigned l8 = min(run_from(m8, n8, lane), room);
            const unsigned l16 = min(run_from(m16, n16, lane), room);
            if (l16 > best_len) { best_len = l16; best_off = 16; }
            if (l8 >= best_len && l8 >= 4) { best_len = l8; best_off = 8; }
        }
        m8 = n8;
        m16 = n16;

        // ---- greedy selection on the scalar unit ----
        const unsigned long long cand_mask = __ballot(in_range && best_len >= 4u);
        unsigned long long sel = 0, covered = skip >= 64u ? ~0ull : ((1ull << skip) - 1ull);
        unsigned cursor = min(skip, 64u);
        unsigned carry = skip >= 64u ? skip - 64u : 0u;
        while (cursor < 64u) {
            const unsigned long long rest = cand_mask >> cursor;
            if (!rest)
                break;
            const unsigned s = cursor + (unsigned)__builtin_ctzll(rest);
            const unsigned len = (unsigned)__builtin_amdgcn_readlane((int)best_len, (int)s);
            sel |= 1ull << s;
            const unsigned e = s + len;
            covered |= (e >= 64u ? ~0ull : ((1ull << e) - 1ull)) & ~((1ull << s) - 1ull);
            cursor = e;
            if (e >= 64u) {
                carry = e - 64u;
                break;
            }
        }
        skip = carry;
        sel = uniform64(sel);
        covered = uniform64(covered);

        // ---- emission ----
        const unsigned long long valid = n - t * 64u >= 64u ? ~0ull : ((1ull << (n - t * 64u)) - 1ull);
        const unsigned long long lit = ~covered & valid;
        const unsigned long long starts = lit & ~(lit << 1);
        const bool is_lit = (lit >> lane) & 1ull;
        const bool is_start = (starts >> lane) & 1ull;
        const bool is_copy = (sel >> lane) & 1ull;
        unsigned run = 0;
        if (is_start) {
            const unsigned long long a = ~(lit >> lane);
            run = a ? (unsigned)__builtin_ctzll(a) : 64u;
        }
        const bool copy1 = best_len < 12u && best_off < 2048u;
        unsigned emit = 0;
        if (is_lit)
            emit = 1u + (is_start ? (run > 60u ? 2u : 1u) : 0u);
        else if (is_copy)
            emit = copy1 ? 2u : 3u;
        const unsigned long long e0 = __ballot(emit & 1u), e1 = __ballot(emit & 2u);
        unsigned at = out_pos + bits_below(e0) + 2u * bits_below(e1);
        if (is_lit) {
            if (is_start) {
                if (run > 60u) {
                    out[at++] = (uint8_t)(60u << 2);
                    out[at++] = (uint8_t)(run - 1u);
                } else {
                    out[at++] = (uint8_t)((run - 1u) << 2);
                }
            }
            out[at] = data[p];
        } else if (is_copy) {
            if (copy1) {
                out[at] = (uint8_t)(1u | ((best_len - 4u) << 2) | ((best_off >> 8) << 5));
                out[at + 1] = (uint8_t)best_off;
            } else {
                out[at] = (uint8_t)(2u | ((best_len - 1u) << 2));
                out[at + 1] = (uint8_t)best_off;
                out[at + 2] = (uint8_t)(best_off >> 8);
            }
        }
        out_pos += (unsigned)__popcll(e0) + 2u * (unsigned)__popcll(e1);
        // remember only positions where an element starts (as libsnappy does): bytes inside a
        // copy would otherwise evict the older, still useful entries of a small table
        if ((is_lit || is_copy) && my_hash != 0xFFFFFFFFu)
            table[my_hash] = (uint16_t)p;
    }
    if (lane == 0)
        frag_sizes[f] = out_pos;
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
//   * match extension compares 16 bytes per step (<= 4 steps), the covered-by-a-copy mask comes
//     from a DPP max-scan instead of 64-bit scalar arithmetic in the selection loop.

constexpr unsigned kWgWaves = 4;
constexpr int kFixed = 4;      // candidates at 1..4 block pitches (8-byte or 16-byte blocks)
constexpr unsigned kWgHashBits = 12;
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

// run of consecutive 1 bits starting at bit `lane` of next:cur, capped at 32 (enough when a lane is
// 2 bytes and copies are at most 64 bytes): one 32-bit funnel shift instead of 64-bit shifts
__device__ __forceinline__ unsigned run_from32(unsigned long long cur, unsigned long long next, unsigned lane)
{
    const unsigned lo = lane < 32u ? (unsigned)cur : (unsigned)(cur >> 32);
    const unsigned hi = lane < 32u ? (unsigned)(cur >> 32) : (unsigned)next;
    const unsigned w = __builtin_amdgcn_alignbit(hi, lo, lane & 31u);
    const unsigned inv = ~w;
    return inv ? (unsigned)__builtin_ctz(inv) : 32u;
}

// equal bytes of data[a..] and data[b..], 16 per step, at most `limit`
__device__ __forceinline__ unsigned match_extend16(const uint32_t *dw, unsigned a, unsigned b, unsigned limit)
{
    unsigned l = 0;
    while (l < limit) {
        const unsigned wa = (a + l) >> 2, sa = (a + l) & 3u, wb = (b + l) >> 2, sb = (b + l) & 3u;
        const unsigned a0 = dw[wa], a1 = dw[wa + 1], a2 = dw[wa + 2], a3 = dw[wa + 3], a4 = dw[wa + 4];
        const unsigned b0 = dw[wb], b1 = dw[wb + 1], b2 = dw[wb + 2], b3 = dw[wb + 3], b4 = dw[wb + 4];
        const unsigned d0 = __builtin_amdgcn_alignbyte(a1, a0, sa) ^ __builtin_amdgcn_alignbyte(b1, b0, sb);
        const unsigned d1 = __builtin_amdgcn_alignbyte(a2, a1, sa) ^ __builtin_amdgcn_alignbyte(b2, b1, sb);
        const unsigned d2 = __builtin_amdgcn_alignbyte(a3, a2, sa) ^ __builtin_amdgcn_alignbyte(b3, b2, sb);
        const unsigned d3 = __builtin_amdgcn_alignbyte(a4, a3, sa) ^ __builtin_amdgcn_alignbyte(b4, b3, sb);
        if (d0 | d1 | d2 | d3) {
            l += d0 ? ((unsigned)__builtin_ctz(d0) >> 3)
               : d1 ? 4u + ((unsigned)__builtin_ctz(d1) >> 3)
               : d2 ? 8u + ((unsigned)__builtin_ctz(d2) >> 3)
                    : 12u + ((unsigned)__builtin_ctz(d3) >> 3);
            break;
        }
        l += 16u;
    }
    return min(l, limit);
}

template <unsigned GRAN>     // bytes per lane: 1, 2 or 4 (positions, offsets and lengths all multiples of GRAN)
__global__ __launch_bounds__(64 * kWgWaves) void snappy_compress_wg_kernel(const HapGpuFrameEnc *__restrict__ frames,
                                                                unsigned frag_log2, uint8_t *__restrict__ slots,
                                                                unsigned slot_stride, uint32_t *__restrict__ frag_sizes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const unsigned frag_bytes = 1u << frag_log2;
    uint32_t *dataw = reinterpret_cast<uint32_t *>(smem);                        // frag_bytes + 32
    const uint8_t *data = smem;
    uint32_t *table = reinterpret_cast<uint32_t *>(smem + frag_bytes + 32);      // kWgHashEntries
    uint32_t *roundsz = table + kWgHashEntries;                                  // kWgWaves

    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const HapGpuFrameEnc &frame = frames[blockIdx.z];
    if (blockIdx.y >= frame.tex_count)
        return;
    const HapGpuTexEnc &tex = frame.tex[blockIdx.y];
    if (tex.compressor != 1u || (1u << tex.reserved) != GRAN)
        return;
    const unsigned x = blockIdx.x;
    if (x >= tex.chunk_count * tex.frags_per_chunk)
        return;
    const unsigned chunk = x / tex.frags_per_chunk, j = x - chunk * tex.frags_per_chunk;
    const unsigned begin = j << frag_log2;
    const unsigned n = min(frag_bytes, tex.chunk_bytes - begin);
    const uint8_t *src = (const uint8_t *)tex.src + (size_t)chunk * tex.chunk_bytes + begin;
    const unsigned f = tex.frag_first + x;
    uint8_t *out = slots + (size_t)f * slot_stride;

    if (((uintptr_t)src & 15u) == 0) {
        for (unsigned i = tid * 16u; i < n + 32u; i += 1024u * kWgWaves) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i + 16u <= n) {
                v = *reinterpret_cast<const uint4 *>(src + i);
            } else if (i < n) {
                unsigned w[4] = {0, 0, 0, 0};
                for (unsigned k = 0; i + k < n; k++)
                    w[k >> 2] |= (unsigned)src[i + k] << (8 * (k & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4 *>(smem + i) = v;
        }
    } else {
        for (unsigned i = tid; i < n + 32u; i += 64u * kWgWaves)
            smem[i] = i < n ? src[i] : (uint8_t)0;
    }
    for (unsigned i = tid; i < kWgHashEntries; i += 64u * kWgWaves)
        table[i] = 0u;
    __syncthreads();

    constexpr unsigned TB = 64u * GRAN;                 // bytes per tile
    const uint16_t *data16 = reinterpret_cast<const uint16_t *>(smem);
    const uint32_t *data32 = reinterpret_cast<const uint32_t *>(smem);
    const unsigned tiles = (n + TB - 1u) / TB, supers = (tiles + 1u) / 2u;
    unsigned round_base = 0;
    // DXT1 / RGTC1 textures are arrays of 8-byte blocks, everything else 16-byte blocks (hap.c:287-294)
    const unsigned pitch = (tex.format_nibble == 0xBu || tex.format_nibble == 0x1u) ? 8u : 16u;

    for (unsigned base = 0; base < supers; base += kWgWaves) {
        const unsigned k = base + wave;
        const bool have = k < supers;
        // per-tile emission plan, kept in registers across the round barrier
        unsigned p_len[2] = {0, 0}, p_off[2] = {0, 0}, p_hash[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        unsigned p_at[2] = {0, 0}, p_run[2] = {0, 0}, p_flags[2] = {0, 0}, p_byte[2] = {0, 0};
        (void)data16;
        (void)data32;
        unsigned total = 0;
        if (have) {
            const unsigned super_end = min(n, (2u * k + 2u) * TB);
            // equality ballots for the fixed distances (block pitches of DXT data)
            unsigned long long eq[kFixed][2];
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const unsigned p = (2u * k + sub) * TB + GRAN * lane;
                const bool in = p < n;
                const unsigned here = GRAN == 4 ? data32[p >> 2] : GRAN == 2 ? (unsigned)data16[p >> 1] : (unsigned)data[p];
#pragma unroll
                for (int d = 0; d < kFixed; d++) {
                    const unsigned dist = (unsigned)(d + 1) * pitch;
                    const unsigned back = p >= dist ? p - dist : 0u;
                    const unsigned there = GRAN == 4 ? data32[back >> 2] : GRAN == 2 ? (unsigned)data16[back >> 1] : (unsigned)data[back];
                    eq[d][sub] = __ballot(in && p >= dist && here == there);
                }
            }
            unsigned skip = 0;
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const unsigned p = (2u * k + sub) * TB + GRAN * lane;
                const bool in_range = p < n;
                const unsigned room = in_range ? (min(64u, super_end - p) & ~(GRAN - 1u)) : 0u;
                unsigned best_len = 0, best_off = 0, my_hash = 0xFFFFFFFFu;
                const unsigned cur = lds_load32(dataw, p);
                if (p + 4u <= n) {
#ifdef HAP_MUL24_HASH
                    const unsigned h = (__umul24(cur & 0xFFFFFFu, 0x9E3779u) + __umul24(cur >> 8, 0x85EBCBu)) >> (32u - kWgHashBits);
#else
                    const unsigned h = (cur * 0x1e35a7bdu) >> (32u - kWgHashBits);
#endif
                    const unsigned cand = table[h];
                    my_hash = h;
#ifdef HAP_NO_HASH
                    if (false) {
#else
                    if (cand < p && room >= 4u && lds_load32(dataw, cand) == cur) {
#endif
                        best_len = (4u + match_extend16(dataw, cand + 4u, p + 4u, room - 4u)) & ~(GRAN - 1u);
                        best_off = p - cand;
                    }
                }
#pragma unroll
                for (int d = kFixed - 1; d >= 0; d--) {          // nearer distances win ties
                    const unsigned long long nx = sub == 0 ? eq[d][1] : 0ull;
                    const unsigned l = min(GRAN >= 2 ? GRAN * run_from32(eq[d][sub], nx, lane)
                                                     : run_from(eq[d][sub], nx, lane), room);
                    if (l >= best_len && l >= 4u) { best_len = l; best_off = (unsigned)(d + 1) * pitch; }
                }
                // greedy selection: the scalar unit hops from chosen copy to chosen copy
#ifdef HAP_MIN_COPY2      /* experiment: 3-byte copies only from this length up (-1.5 % decode time, +0.3 % bytes) */
                const unsigned long long cand_mask = __ballot(in_range && best_len >= 4u &&
                                                              (best_len >= HAP_MIN_COPY2 || best_off < 2048u));
#else
                const unsigned long long cand_mask = __ballot(in_range && best_len >= 4u);
#endif
                unsigned long long sel = 0;
                unsigned cursor = min(skip, 64u);
                unsigned carry = skip > 64u ? skip - 64u : 0u;
                while (cursor < 64u) {
                    const unsigned long long rest = cand_mask >> cursor;
                    if (!rest)
                        break;
                    const unsigned s = cursor + (unsigned)__builtin_ctzll(rest);
                    sel |= 1ull << s;
                    cursor = s + (unsigned)__builtin_amdgcn_readlane((int)best_len, (int)s) / GRAN;
                    if (cursor > 64u)
                        carry = cursor - 64u;
                }
                const bool is_copy = (sel >> lane) & 1ull;
                // covered[l] <=> some chosen copy (or the carry-in) spans position l
                const int reach = cwave_scan_max(is_copy ? (int)(lane + best_len / GRAN) : 0);
                const bool covered = (unsigned)reach > lane || lane < skip;
                skip = carry;
                const unsigned long long lit = __ballot(in_range && !covered);
                const unsigned long long starts = lit & ~(lit << 1);
                const bool is_lit = (lit >> lane) & 1ull;
                const bool is_start = (starts >> lane) & 1ull;
                unsigned run = 0;                          // literal run length in BYTES
                if (is_start) {
                    const unsigned long long a = ~(lit >> lane);
                    run = GRAN * (a ? (unsigned)__builtin_ctzll(a) : 64u);
                }
                const bool copy1 = best_len < 12u && best_off < 2048u;
                unsigned emit = 0;
                if (is_lit)
                    emit = GRAN + (is_start ? (run > 60u ? 2u : 1u) : 0u);
                else if (is_copy)
                    emit = copy1 ? 2u : 3u;
                const unsigned long long e0 = __ballot(emit & 1u), e1 = __ballot(emit & 2u);
                const unsigned long long e2 = GRAN >= 2 ? __ballot(emit & 4u) : 0ull;
                p_at[sub] = total + bits_below(e0) + 2u * bits_below(e1) + (GRAN >= 2 ? 4u * bits_below(e2) : 0u);
                total += (unsigned)__popcll(e0) + 2u * (unsigned)__popcll(e1) + 4u * (unsigned)__popcll(e2);
                p_len[sub] = best_len;
                p_off[sub] = best_off;
                p_hash[sub] = my_hash;
                p_run[sub] = run;
                p_byte[sub] = GRAN == 4 ? cur : (cur & (GRAN == 2 ? 0xFFFFu : 0xFFu));
                p_flags[sub] = (is_lit ? 1u : 0u) | (is_start ? 2u : 0u) | (is_copy ? 4u : 0u) | (copy1 ? 8u : 0u);
            }
        }
        if (lane == 0)
            roundsz[wave] = total;
        __syncthreads();
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
            for (int sub = 0; sub < 2; sub++) {
                const unsigned p = (2u * k + sub) * TB + GRAN * lane;
                unsigned at = my_base + p_at[sub];
                const unsigned fl = p_flags[sub];
                if (fl & 1u) {
                    if (fl & 2u) {
                        if (p_run[sub] > 60u) {
                            out[at++] = (uint8_t)(60u << 2);
                            out[at++] = (uint8_t)(p_run[sub] - 1u);
                        } else {
                            out[at++] = (uint8_t)((p_run[sub] - 1u) << 2);
                        }
                    }
                    out[at] = (uint8_t)p_byte[sub];
                    if (GRAN >= 2)
                        out[at + 1] = (uint8_t)(p_byte[sub] >> 8);
                    if (GRAN == 4) {
                        out[at + 2] = (uint8_t)(p_byte[sub] >> 16);
                        out[at + 3] = (uint8_t)(p_byte[sub] >> 24);
                    }
                } else if (fl & 4u) {
                    if (fl & 8u) {
                        out[at] = (uint8_t)(1u | ((p_len[sub] - 4u) << 2) | ((p_off[sub] >> 8) << 5));
                        out[at + 1] = (uint8_t)p_off[sub];
                    } else {
                        out[at] = (uint8_t)(2u | ((p_len[sub] - 1u) << 2));
                        out[at + 1] = (uint8_t)p_off[sub];
                        out[at + 2] = (uint8_t)(p_off[sub] >> 8);
                    }
                }
                // only element starts are remembered (see the single-wave kernel)
                if ((fl & 5u) && p_hash[sub] != 0xFFFFFFFFu)
                    atomicMax(&table[p_hash[sub]], p);
            }
        }
        __syncthreads();
    }
    if (tid == 0)
        frag_sizes[f] = round_base;
}

} // namespace

// LDS bytes needed per workgroup for a fragment size
static unsigned compress_lds_bytes(unsigned frag_log2) { return (1u << frag_log2) + 16u + kHashEntries * 2u; }

extern "C" int hapgpu_launch_snappy_compress(const HapGpuFrameEnc *frames, unsigned frame_count,
                                             unsigned max_frags_per_texture, unsigned frag_log2, void *slots,
                                             unsigned slot_stride, uint32_t *frag_sizes, unsigned granularity_mask,
                                             hipStream_t stream)
{
    if (frame_count == 0 || max_frags_per_texture == 0)
        return 0;
    if (frag_log2 < 10 || frag_log2 > 16)
        return 1;
    static const bool use_v1 = getenv("HAP_AMD_COMPRESS_V1") != nullptr;
    if (!use_v1) {
        const unsigned lds2 = (1u << frag_log2) + 32u + kWgHashEntries * 4u + kWgWaves * 4u;
        if (lds2 > 65536u) {
            static bool once2 = false;
            if (!once2) {
                if (hipFuncSetAttribute((const void *)snappy_compress_wg_kernel<1u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
                    hipFuncSetAttribute((const void *)snappy_compress_wg_kernel<2u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
                    hipFuncSetAttribute((const void *)snappy_compress_wg_kernel<4u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
                    return 4;
                once2 = true;
            }
        }
        const dim3 grid(max_frags_per_texture, 2, frame_count);
        if (granularity_mask & 1u)
            hipLaunchKernelGGL(snappy_compress_wg_kernel<1u>, grid, dim3(64 * kWgWaves), lds2, stream, frames, frag_log2, (uint8_t *)slots, slot_stride, frag_sizes);
        if (granularity_mask & 2u)
            hipLaunchKernelGGL(snappy_compress_wg_kernel<2u>, grid, dim3(64 * kWgWaves), lds2, stream, frames, frag_log2, (uint8_t *)slots, slot_stride, frag_sizes);
        if (granularity_mask & 4u)
            hipLaunchKernelGGL(snappy_compress_wg_kernel<4u>, grid, dim3(64 * kWgWaves), lds2, stream, frames, frag_log2, (uint8_t *)slots, slot_stride, frag_sizes);
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
    const unsigned lds = compress_lds_bytes(frag_log2);
    if (lds > 65536u) {
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute((const void *)snappy_compress_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return 4;
            once = true;
        }
    }
    hipLaunchKernelGGL(snappy_compress_kernel, dim3(max_frags_per_texture, 2, frame_count), dim3(64), lds, stream,
                       frames, frag_log2, (uint8_t *)slots, slot_stride, frag_sizes);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
