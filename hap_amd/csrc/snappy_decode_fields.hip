// snappy_decode_fields.hip -- block-per-lane Snappy decoder for "field streams" (gfx950).
//
// Replaces hap_decode_chunk's snappy_uncompress (reference hap.c:606-642, call at hap.c:612) for the frames this
// library writes itself with the fragment table version 4 (private section 0x46, include/hap_gpu.h).  Such a chunk
// is an ordinary Snappy stream -- the reference decodes it unchanged -- whose elements obey extra rules that the
// table announces and this kernel VERIFIES while it parses (any violation fails the unit with
// HAPGPU_STATUS_INDEX_MISMATCH and the host decodes the frame again through the generic kernels):
//
//   * the stream is cut into independent 8 KiB fragments (compressed size of each in the table);
//   * the table holds, per fragment, the compressed bytes AND the output bytes of 64 GROUPS of its elements -- the
//     elements in stream order, ceil(N / 64) to a group (24 bits per group) -- and N;
//   * inside a fragment no element crosses a 128-byte "half-tile" of output;
//   * every element starts and ends on a block FIELD boundary -- DXT5 / YCoCg-DXT5 blocks are 2 + 6 + 4 + 4 bytes
//     (alpha endpoints, alpha indices, colour endpoints, colour indices), DXT1 blocks 4 + 4, RGTC1 blocks 2 + 6 -- and
//     every copy offset is a whole number of blocks.  So each field of each block is produced by exactly one element,
//     either from literal bytes or from the SAME field of an earlier block.
//
// One wavefront decodes one fragment in two phases:
//
//   1. PARSE, one lane per group: the only serial dependency of Snappy, the element chain, is 64 independent chains of
//      EQUAL length (r03's first table gave a lane to every half-tile: the busiest of 64 set the trip count, 21
//      elements against a mean of 6).  Two scans over the table's entries give every group its input position, its
//      output position and its record slots (table version 3 listed the compressed bytes only: a first walk over the
//      tags had to measure every group).  The walk leaves per element a 32-bit record {where its bytes come from,
//      relative to its place in the step | 4 x block distance} and ORs one bit into the start mask of the element's
//      half-tile.
//   2. PRODUCE, one lane per BLOCK, 64 blocks (1 KiB of DXT5) per step: for each of its 4 fields the lane finds the
//      owning element with one popcount of the start mask, reads the record, and turns it into a source descriptor
//      with three instructions (shift-add, subtract, minimum):
//      literal bytes in the staged input, or the same field of block (b - distance) in the output ring.  Sources
//      inside the current step are resolved first by DPP hops (lanes 1, 2, 4, 8 below), then by pointer doubling on
//      lane indices (ds_bpermute, <= 6 rounds); then 4 field reads, one 16-byte LDS store (later steps copy from
//      it) and one 16-byte global store.
//
// All layouts run in ONE launch (the two textures of a Hap Q Alpha frame are units of one array; the layout is uniform
// per wave).
// LDS: the output ring (8 KiB) and the staged compressed bytes share one buffer -- the input is parked at its END,
// where output written by step s never reaches the compressed bytes that later steps still read (checked per group).
// The element records live in the same buffer too, BELOW the parked input, where output only arrives after every
// record has been read; fragments whose records do not fit there -- hardly compressible ones -- keep them in the
// first bytes of their own output range in memory until production overwrites them.
// 9.9 KiB per wave, 16 waves per CU.
// HBM traffic: compressed bytes read once, output written once (algorithmic bytes b(1 + c)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "hapgpu_abi.h"
#include "measurement_guard.h"

namespace {

constexpr unsigned kFragBytes = 8192u;
constexpr unsigned kHalf = 128u;                       // bytes of output per half-tile
constexpr unsigned kHalves = kFragBytes / kHalf;       // 64: one parse lane each
constexpr unsigned kMaxFragCompressed = kFragBytes + 320u;
constexpr int kGuessLdsBytes = 0;          // (see hapgpu_launch_guess_group_tables)
// Switches of the measurement builds (tools/build_variants.sh, which defines HAP_MEASUREMENT_BUILD and writes to
// hap_amd/variants/; measurement_guard.h refuses them in any other build): LDS per wave up or down (occupancy
// studies), the set of DPP hops, one layout's code alone.  None changes what the kernel writes.  (The ablations of
// round 4 that broke the output on purpose -- no rounds, no ring store, no overrun check -- are gone from the source.)
// ring + parked input (at the end of the buffer: see the S computation below) + alignment slack
// (The least an honest stream needs is 16 + 8320 + alignment: with 8416 + 32 bytes buffer + masks + offsets are 9216
// bytes and a CU holds 17 wavefronts instead of 16 -- tried late in r05: C4 decode 0.640 -> 0.651 ms, C5 0.518 -> 0.528,
// SLOWER; fragments that compress badly keep their records in memory sooner, and one more wavefront buys nothing.)
#ifndef SDF_BUF_BYTES
#define SDF_BUF_BYTES (9344u + 32u)
#endif
#ifndef SDF_DYN_LDS
#define SDF_DYN_LDS 0
#endif
#ifndef SDF_HOPS
#define SDF_HOPS 15u
#endif
constexpr unsigned kBufBytes = SDF_BUF_BYTES;

__device__ __forceinline__ int fdpp_shr(int v, int n)
{
    switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    }
}

__device__ __forceinline__ int fwave_scan_add(int v)       // inclusive
{
    v += fdpp_shr(v, 1);
    v += fdpp_shr(v, 2);
    v += fdpp_shr(v, 4);
    v += fdpp_shr(v, 8);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// Block layouts ("fields per block" nibble of the fragment table, include/hap_gpu.h):
//   4: 16-byte blocks of 2 + 6 + 4 + 4 bytes (DXT5, YCoCg-DXT5);  2: 8-byte blocks of 4 + 4 (DXT1);
//   6: 8-byte blocks of 2 + 6 (RGTC1);  8: 16-byte blocks of 4 + 4 + 4 + 4 (opaque formats: BC7, BC6H)
template <unsigned LAYOUT> struct layout_of;
template <> struct layout_of<4u> { static constexpr unsigned fields = 4u, block = 16u, pos_shift = 1u; };
template <> struct layout_of<2u> { static constexpr unsigned fields = 2u, block = 8u, pos_shift = 2u; };
template <> struct layout_of<6u> { static constexpr unsigned fields = 2u, block = 8u, pos_shift = 1u; };
template <> struct layout_of<8u> { static constexpr unsigned fields = 4u, block = 16u, pos_shift = 2u; };

// byte offset of field k inside a block
template <unsigned LAYOUT>
__device__ __forceinline__ constexpr unsigned field_pos(unsigned k)
{
    return LAYOUT == 4u ? (k == 0u ? 0u : k == 1u ? 2u : k == 2u ? 8u : 12u) : (LAYOUT == 2u || LAYOUT == 8u) ? 4u * k : 2u * k;
}

// positions (in units of 1 << pos_shift bytes) at which an element may start, as a mask over one 32-bit word of the
// start mask: 16-byte blocks are 8 positions with fields at 0, 1, 4, 6; [2, 6] blocks 4 positions with fields at 0, 1
template <unsigned LAYOUT>
__device__ __forceinline__ constexpr unsigned start_positions() { return LAYOUT == 4u ? 0x53535353u : LAYOUT == 6u ? 0x33333333u : 0xFFFFFFFFu; }

// waits for every outstanding LDS operation of the wave (one wait for a batch of reads instead of one per use)
__device__ __forceinline__ void lds_wait() { __builtin_amdgcn_s_waitcnt(0xC07F); }     // lgkmcnt(0), vmcnt / expcnt untouched

// Fails the unit: the host decodes the frame again without the table (generic kernels).
__device__ __forceinline__ void fail_unit(HapGpuDecodeJob *job, unsigned lane)
{
    if (lane == 0)
        atomicCAS(&job->status, 0u, HAPGPU_STATUS_INDEX_MISMATCH);
}

// Everything in memory is reached through global-address-space pointers: the addresses come out of the unit as
// integers, and a pointer of unknown address space would be accessed with FLAT instructions -- 64-bit address
// arithmetic per access, and waits that count against the LDS counter as well.
typedef const uint8_t __attribute__((address_space(1))) *gin_t;
typedef uint8_t __attribute__((address_space(1))) *gout_t;
struct alignas(16) quad16 { uint32_t a, b, c, d; };
struct alignas(8) pair8 { uint32_t a, b; };
__device__ __forceinline__ uint4 gload16(gin_t p)
{
    const quad16 __attribute__((address_space(1))) *q = reinterpret_cast<const quad16 __attribute__((address_space(1))) *>(p);
    return make_uint4(q->a, q->b, q->c, q->d);
}
__device__ __forceinline__ void gstore16(gout_t p, const uint4 v)
{
    quad16 __attribute__((address_space(1))) *q = reinterpret_cast<quad16 __attribute__((address_space(1))) *>(p);
    q->a = v.x; q->b = v.y; q->c = v.z; q->d = v.w;
}
__device__ __forceinline__ void gstore8(gout_t p, const uint2 v)
{
    pair8 __attribute__((address_space(1))) *q = reinterpret_cast<pair8 __attribute__((address_space(1))) *>(p);
    q->a = v.x; q->b = v.y;
}

// 16 bytes of the unit's input at aligned coordinate x (coordinates are relative to src - shift); bytes outside
// [shift, in_end) read as zero and are never touched in memory
// (bytes below `shift` belong to the same frame -- its headers precede every chunk -- and are fetched with the piece;
// `readable_end` = in_end + the bytes known to follow the fragment inside the texture section)
__device__ __forceinline__ uint4 load_input16(gin_t src_al, unsigned x, unsigned shift, unsigned in_end, unsigned readable_end)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (x < in_end && x + 16u <= readable_end) {
        v = gload16(src_al + x);
    } else if (x < in_end) {
        unsigned w[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (unsigned k = 0; k < 16u; k++) {
            const unsigned y = x + k;
            if (y >= shift && y < in_end)
                w[k >> 2] |= (unsigned)src_al[y] << (8u * (k & 3u));
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

template <unsigned LAYOUT>
__device__ __forceinline__ void decode_fields_unit(const HapGpuDecodeUnit &u, HapGpuDecodeJob *jobs, uint8_t *buf, uint2 *masks,
                                                   uint32_t *coffs, const unsigned lane)
{
    constexpr unsigned PERIOD = layout_of<LAYOUT>::fields;
    constexpr unsigned kBlock = layout_of<LAYOUT>::block;
    constexpr unsigned kBlocksPerHalf = kHalf / kBlock;            // 8 or 16
    constexpr unsigned kStepBytes = 64u * kBlock;                  // 1024 or 512
    constexpr unsigned kHalvesPerStep = kStepBytes / kHalf;        // 8 or 4
    constexpr unsigned kPosShift = layout_of<LAYOUT>::pos_shift;    // element start positions are kept in 2- / 4-byte units
    const uint32_t *bufw = reinterpret_cast<const uint32_t *>(buf);
    HapGpuDecodeJob *job = &jobs[u.job];
    const gin_t src = (gin_t)u.src;
    const gout_t dst = (gout_t)u.dst;
    const gin_t group_table = (gin_t)u.aux;
    const unsigned total = u.src_len, out_len = u.dst_len;
    if (out_len == 0u || out_len > kFragBytes || (out_len % kBlock) != 0u || total > kMaxFragCompressed || !group_table) {
        fail_unit(job, lane);
        return;
    }
    const unsigned nhalf = (out_len + kHalf - 1u) / kHalf;

    // everything the unit needs from memory is requested at once: the job's status word, the group table and
    // the first 4 KiB of input (what comes later is fetched by the loop below)
    const unsigned shift = (unsigned)((uintptr_t)src & 15u);
    const gin_t src_al = src - shift;
    const unsigned in_end = shift + total;
    const unsigned readable_end = in_end + (unsigned)(u.reserved & 15u);
    const unsigned job_status = __builtin_nontemporal_load(&job->status);
    // fragment table version 4: 64 groups x 24 bits little endian (compressed bytes | bytes produced << 12), then the
    // number of elements (LE16)
    // (one dword at any byte address per lane -- lane 63's reaches into the element count, inside the table -- and one 16-bit load)
    struct __attribute__((packed)) any32 { uint32_t v; };
    struct __attribute__((packed)) any16 { uint16_t v; };
    const unsigned gentry = reinterpret_cast<const any32 __attribute__((address_space(1))) *>(group_table + 3u * lane)->v & 0xFFFFFFu;
    const unsigned elements_raw = reinterpret_cast<const any16 __attribute__((address_space(1))) *>(group_table + 192)->v;
    uint4 early[4];
#pragma unroll
    for (unsigned i = 0; i < 4u; i++)
        early[i] = load_input16(src_al, i * 1024u + lane * 16u, shift, in_end, readable_end);
    if (job_status != 0u)
        return;

    // ---- group table -> where every lane starts reading, where its output begins, which record slots are its own ----
    // (the table of version 3 listed the compressed bytes only, and a first walk over the tags measured every group:
    // 24 instructions a turn, 7 % of the kernel)
    const unsigned gsz = gentry & 0xFFFu, gout = gentry >> 12;
    const unsigned gincl = (unsigned)fwave_scan_add((int)gsz);
    const unsigned oincl = (unsigned)fwave_scan_add((int)gout);
    const unsigned coff = gincl - gsz;
    const unsigned obegin = oincl - gout;                          // output position of the group's first element
    const unsigned elements = (unsigned)__builtin_amdgcn_readfirstlane((int)elements_raw);
    // every element is at least two bytes (a field) and every field belongs to one element: at most out_len / 4 of them
    // (the records, 4 bytes each, then fit the unit's own output range if they have to go to memory)
    if ((unsigned)__builtin_amdgcn_readlane((int)gincl, 63) != total || (unsigned)__builtin_amdgcn_readlane((int)oincl, 63) != out_len ||
        elements == 0u || 4u * elements > out_len) {
        fail_unit(job, lane);
        return;
    }
    const unsigned G = (elements + 63u) >> 6;                      // elements per group (the last groups: fewer, or none)
    const unsigned rbase = min(lane * G, elements);                // ordinal of the group's first element
    const unsigned count_g = min(rbase + G, elements) - rbase;
    // The input is parked at the END of the buffer: output written by the steps before step s then never reaches the
    // compressed bytes that step s reads as literals, as long as what is left of the input at any point fits between the
    // output produced so far and the end of the buffer -- true for every honest stream (<= 130 bytes per half-tile),
    // and verified element by element in the walk below.
    const unsigned S = ((kBufBytes - 16u - total - shift) & ~15u) + shift;                 // S = shift (mod 16)
    {
        // (every lane stores only granules that hold input)
        uint8_t *park = buf + (S - shift) + lane * 16u;
#pragma unroll
        for (unsigned i = 0; i < 4u; i++)
            if (i * 1024u + lane * 16u < in_end)
                *reinterpret_cast<uint4 *>(park + i * 1024u) = early[i];
        for (unsigned x = 4096u; x < in_end; x += 1024u)
            if (x + lane * 16u < in_end)
                *reinterpret_cast<uint4 *>(park + x) = load_input16(src_al, x + lane * 16u, shift, in_end, readable_end);
    }
    masks[lane] = make_uint2(0u, 0u);
    __syncthreads();

    // ---- 1. parse: lane g walks the elements of group g -- every group holds the same number of them ----
    const unsigned cbegin = S + coff, cend = cbegin + gsz;
    // records (4 bytes each) below the parked input when they fit, else in the unit's own output range (4-byte aligned) in memory
    const unsigned rbytes = 4u * elements;
    const bool rec_in_lds = rbytes <= S - shift;
    const gout_t rec_mem = (gout_t)(((uintptr_t)dst + 3u) & ~(uintptr_t)3u);
    if (!rec_in_lds && rbytes + (unsigned)(((uintptr_t)rec_mem - (uintptr_t)dst)) > out_len) {
        fail_unit(job, lane);
        return;
    }

    // The walk.  Straight-line code under the loop's exec mask: the three element kinds are decoded side by side and
    // selected.  Per element one 32-bit record -- everything a field's descriptor needs from its element, formed ONCE
    // per element here instead of once per field in the lookup below (5.5 turns against 32 columns) --
    //     bits 23..0:  literal: (address of its bytes in the parked input) - (its position in the step)
    //                  copy:    (start of the step in the ring) - (copy offset)
    //                  -- add the field's position in the step and either is the address the field's bytes come from
    //     bits 31..24: copy: 4 x distance in blocks where that is below 64 (the source may lie in the same step);
    //                  255 otherwise and for literals
    // and one bit of its half-tile's start mask (bit = position in 2- or 4-byte units).  Promise checks are accumulated
    // and looked at once at the end.
    constexpr unsigned kRecShift = kBlock == 16u ? 2u : 1u;       // byte offset -> 4 x blocks
    {
        unsigned cp = cbegin;
        unsigned p = obegin;                               // output position inside the fragment
        unsigned recp = 4u * rbase;
        unsigned acc_or = 0;                               // OR of copy offsets (low bits) and start positions << 17
        unsigned max_kind = 0;                             // 3 = a copy-4 element
        int max_reach = 0;                                 // how far before the fragment the farthest copy reaches
        unsigned min_off = 0xFFFFFu, max_up = 0;           // smallest copy offset, largest literal length code
        unsigned crossed = 0;                              // an element that leaves its half-tile
        unsigned overrun = 0;                              // an element whose bytes the output of earlier steps would reach
        uint32_t *const mask_words = reinterpret_cast<uint32_t *>(masks);
        // (two typed pointers and a uniform branch at the store: one pointer chosen between LDS and memory would make
        // every record a flat store)
#pragma unroll 1
        for (unsigned turn = 0; turn < G; turn++) {        // (uniform trip count: the table says how many elements a group has)
            if (turn < count_g) {
                const unsigned aw = cp >> 2;
                const unsigned w = __builtin_amdgcn_alignbyte(bufw[aw + 1u], bufw[aw], cp);     // bytes cp .. cp+3 (shift = cp & 3)
                const unsigned kind = w & 3u, up = __builtin_amdgcn_ubfe(w, 2u, 6u);
                const unsigned b1 = __builtin_amdgcn_ubfe(w, 8u, 8u);
                const bool is_lit = kind == 0u, is_c1 = kind == 1u;
                const bool lng = is_lit && up == 60u;
                const unsigned lngv = lng ? 1u : 0u;
                // length - 1: literal / copy-2: tag >> 2 (long literal: the next byte); copy-1: 3 + 3 bits
                unsigned lm1 = lng ? b1 : up;
                lm1 = is_c1 ? __builtin_amdgcn_ubfe(w, 2u, 3u) + 3u : lm1;
                const unsigned adv = is_lit ? lm1 + lngv + 2u : kind + 1u;
                const unsigned off = is_c1 ? (((w << 3) & 0x700u) | b1) : __builtin_amdgcn_ubfe(w, 8u, 16u);
                const unsigned hp = p & (kHalf - 1u);                           // position inside the half-tile
                // promises: whole blocks back (low offset bits 0), at least one, not before the fragment; no copy-4
                // (kind 3); no literal with 2..4 length bytes (tag >> 2 in 61..63); starts on 2- / 4-byte positions;
                // the element ends inside its half-tile
                const unsigned offx = is_lit ? 0x10000u : off;                  // (a literal counts as offset 64 Ki: neutral below)
                acc_or |= offx | (p << 17);
                min_off = min(min_off, offx);
                max_up = max(max_up, is_lit ? up : 0u);
                max_kind = max(max_kind, kind);
                max_reach = max(max_reach, (int)(offx & 0xFFFFu) - (int)p);     // > 0: a copy from before the fragment
                crossed |= (hp + lm1) >> 7;
                overrun |= (p & ~(kStepBytes - 1u)) > cp ? 1u : 0u;            // the steps before this one write up to there
                {
                    const unsigned from = (is_lit ? cp + 1u + lngv : p - off) - (p & (kStepBytes - 1u));
                    const unsigned record = (from & 0xFFFFFFu) | (min(offx >> kRecShift, 255u) << 24);
                    if (rec_in_lds)
                        *reinterpret_cast<uint32_t *>(buf + recp) = record;
                    else
                        *reinterpret_cast<uint32_t __attribute__((address_space(1))) *>(rec_mem + recp) = record;
                }
                recp += 4u;
                const unsigned bit = hp >> kPosShift;                           // 0 .. 63 (2-byte positions) or 0 .. 31
                // (a table that lies may send p anywhere: the word index is kept inside the masks)
                atomicOr(&mask_words[(2u * (p >> 7) + (bit >> 5)) & (2u * kHalves - 1u)], 1u << (bit & 31u));
                p += lm1 + 1u;
                cp += adv;
            }
        }
        // an element that overshoots its group's bytes, or a group with another number of elements than the table says,
        // leaves cp or p off the mark; starts off a field boundary show in the masks (checked below, per half-tile)
        const bool bad = max_kind == 3u || max_reach > 0 || (acc_or & (kBlock - 1u)) != 0u ||
                         ((acc_or >> 17) & ((1u << kPosShift) - 1u)) != 0u || min_off < kBlock || max_up > 60u ||
                         crossed != 0u || overrun != 0u ||
                         cp != cend || p != obegin + gout;
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
            fail_unit(job, lane);
            return;
        }
    }
    if (!rec_in_lds)
        __threadfence();
    __syncthreads();
    // per half-tile: the start mask may only have bits where fields begin (16-byte blocks: bytes 0, 2, 8, 12), the
    // half-tile begins with an element, and the ordinal of its first element follows from the counts
    {
        const uint2 m = masks[lane];
        const bool live = lane < nhalf;
        const unsigned starts = live ? __builtin_popcount(m.x) + __builtin_popcount(m.y) : 0u;
        const unsigned sincl = (unsigned)fwave_scan_add((int)starts);
        const bool bad = live && (((m.x | m.y) & ~start_positions<LAYOUT>()) != 0u || (m.x & 1u) == 0u);
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull || (unsigned)__builtin_amdgcn_readlane((int)sincl, 63) != elements) {
            fail_unit(job, lane);
            return;
        }
        coffs[lane] = sincl - starts;                       // ordinal of the half-tile's first element
    }
    __syncthreads();

    // ---- 2. produce: lane = block, 64 blocks per step ----
    const bool dst_wide = ((uintptr_t)dst & (kBlock - 1u)) == 0u;
    // per-lane constants: the block's place inside its half-tile, and for each of its fields the mask of start bits
    // at or below it (in the 32-bit half of the start mask that covers the block)
    const unsigned b = lane & (kBlocksPerHalf - 1u);
    const unsigned hsub = lane / kBlocksPerHalf;                           // half-tile of the step
    // (64 positions per half-tile in two words when positions are 2 bytes; 32 four-byte positions fit one word)
    const bool upper = kPosShift == 1u && b >= kBlocksPerHalf / 2u;
    unsigned le[PERIOD];
    int lanec[PERIOD];                                                     // (the field's position in the step) << 8 | 4 x lane
#pragma unroll
    for (unsigned k = 0; k < PERIOD; k++) {
        const unsigned fb = b * kBlock + field_pos<LAYOUT>(k);             // byte position of the field in the half-tile
        const unsigned q = (fb >> kPosShift) & 31u;
        le[k] = (2u << q) - 1u;
        lanec[k] = (int)(((lane * kBlock + field_pos<LAYOUT>(k)) << 8) | (lane * 4u));
    }
    const unsigned lane4s = lane * 4u + 0x80000000u;
    // 2a. every step's fields -> source descriptors ("state"): >= 0 resolved: (an address in buf) << 8 -- literal bytes, or
    //     the ring for a copy whose source block lies in an earlier step -- | 4 x the own lane; < 0 pending: sign bit |
    //     4 x (source lane in the same step).
    //     All steps are looked up before anything is produced: the LDS round trips of the 8 steps overlap.
    //     (Lanes beyond the end of a short fragment compute garbage that nothing reads: sources are always lower lanes.)
    constexpr unsigned kMaxSteps = kFragBytes / kStepBytes;               // 8 or 16
    int state[kMaxSteps][PERIOD];
    {
        // (phases with one wait each: all mask / offset reads, then all record reads, then arithmetic -- instead of a
        // wait in front of every use)
        unsigned mx[kMaxSteps], my[kMaxSteps];
        int cof[kMaxSteps];
#pragma unroll
        for (unsigned s = 0; s < kMaxSteps; s++) {
            const unsigned hh = s * kHalvesPerStep + hsub;
            const uint2 m = masks[hh];
            mx[s] = m.x;
            my[s] = m.y;
            cof[s] = (int)coffs[hh];
        }
        lds_wait();
        unsigned r[kMaxSteps][PERIOD];
        if (rec_in_lds) {
#pragma unroll
            for (unsigned s = 0; s < kMaxSteps; s++) {
                const unsigned msel = upper ? my[s] : mx[s];
                const int ebase = (upper ? (int)__builtin_popcount(mx[s]) - 1 : -1) + cof[s];
                const uint32_t *rb = reinterpret_cast<const uint32_t *>(buf);
#pragma unroll
                for (unsigned k = 0; k < PERIOD; k++) {
                    // ordinal of the element that owns the field (a parsed half-tile always starts with an element: >= 0)
                    const int e = (int)__builtin_popcount(msel & le[k]) + ebase;
                    r[s][k] = rb[e];
                }
                if ((s & 3u) == 3u)
                    lds_wait();                                             // (at most 16 LDS results outstanding)
            }
            lds_wait();
        } else {
            // (the wave's own stores, read back past the CU's L1, which may hold these lines as they were before)
            const uint32_t __attribute__((address_space(1))) *rb = reinterpret_cast<const uint32_t __attribute__((address_space(1))) *>(rec_mem);
#pragma unroll
            for (unsigned s = 0; s < kMaxSteps; s++) {
                const unsigned msel = upper ? my[s] : mx[s];
                const int ebase = (upper ? (int)__builtin_popcount(mx[s]) - 1 : -1) + cof[s];
#pragma unroll
                for (unsigned k = 0; k < PERIOD; k++) {
                    const int e = (int)__builtin_popcount(msel & le[k]) + ebase;
                    r[s][k] = __hip_atomic_load(rb + ((64u * s + lane) * kBlock < out_len ? e : 0), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // Descriptors keep the address in bits 31..8 and a lane number (x 4) in bits 7..0: a pending field names the lane it
        // copies from, a resolved one ITSELF -- so that one ds_bpermute addressed by the descriptor fetches the next
        // descriptor of the chain for pending fields and the same descriptor again for resolved ones (directly, or
        // from the lane at the root of their chain, which holds the same value): no select after the fetch.
        // Three instructions a field: the record's low 24 bits + the field's position = its address if it is a literal or
        // copies from an earlier step; the sign bit | 4 x (lane - distance) if the source block lies in this step --
        // which wraps to a huge positive number when it does not (distance > lane; code 255 for literals): the minimum.
#pragma unroll
        for (unsigned s = 0; s < kMaxSteps; s++) {
#pragma unroll
            for (unsigned k = 0; k < PERIOD; k++) {
                const unsigned rr = r[s][k];
                const int res = (int)(rr << 8) + lanec[k];
                const int pend = (int)(lane4s - (rr >> 24));
                state[s][k] = min(pend, res);
            }
        }
    }
    // 2b. sources produced in the same step: follow the chains to a literal or to an earlier step.
    //     First the hops to a lane 1, 2, 4, 8 below in the same row of 16, taken with DPP moves: runs of copies at
    //     distance 1, 2 or 4 blocks -- the common ones -- collapse to the lane in front of their row.  (Three
    //     instructions a hop, left to the compiler: the two-instruction form -- v_cmp into VCC, v_cndmask_b32_dpp -- was
    //     written out for all 32 columns in r04 and is 3-5 % SLOWER: every pair goes through the one VCC.)
    {
        const unsigned row_lane = lane & 15u;
        // the descriptor "pending, copies from the lane m below" -- or a value no descriptor has, where that lane lies in another row
        const int want1 = row_lane >= 1u ? (int)(0x80000000u + (lane - 1u) * 4u) : 0x40000000;
        const int want2 = row_lane >= 2u ? (int)(0x80000000u + (lane - 2u) * 4u) : 0x40000000;
        const int want4 = row_lane >= 4u ? (int)(0x80000000u + (lane - 4u) * 4u) : 0x40000000;
        const int want8 = row_lane >= 8u ? (int)(0x80000000u + (lane - 8u) * 4u) : 0x40000000;
#pragma unroll
        for (unsigned s = 0; s < kMaxSteps; s++)
#pragma unroll
            for (unsigned k = 0; k < PERIOD; k++) {
                int v = state[s][k];
                int t;
                if (SDF_HOPS & 1u) {
                    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
                    v = v == want1 ? t : v;
                }
                if (SDF_HOPS & 2u) {
                    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
                    v = v == want2 ? t : v;
                }
                if (SDF_HOPS & 4u) {
                    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
                    v = v == want4 ? t : v;
                }
                if (SDF_HOPS & 8u) {
                    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
                    v = v == want8 ? t : v;
                }
                state[s][k] = v;
            }
    }
    //     Then pointer doubling (a chain is at most 63 links long: 6 rounds); the columns of all steps advance together,
    //     so a round is 32 independent ds_bpermutes in flight instead of one dependent LDS round trip per step and
    //     round.  Descriptors are self-addressed (see above): the fetched value IS the new descriptor.  (r04: a round
    //     that fetches only for the columns that still have a pending lane -- a bit per column, one hand-written block
    //     of compare / branch / ds_bpermute per round -- issues a quarter of the ds_bpermutes and is no faster: the
    //     kernel is bound by vector instruction issue, not by the LDS pipe; LABNOTES.md.)
#pragma unroll 1
    for (unsigned round = 0; round < 6u; round++) {
        int any = state[0][0];
#pragma unroll
        for (unsigned s = 0; s < kMaxSteps; s++)
#pragma unroll
            for (unsigned k = 0; k < PERIOD; k++)
                any |= state[s][k];
        // (r05: taking the first round unasked -- hardly a fragment needs none -- made the kernel 1.5 % SLOWER)
        if (__builtin_amdgcn_ballot_w64(any < 0 && lane * kBlock < out_len) == 0ull)
            break;
#pragma unroll
        for (unsigned s = 0; s < kMaxSteps; s++)
#pragma unroll
            for (unsigned k = 0; k < PERIOD; k++)
                state[s][k] = __builtin_amdgcn_ds_bpermute(state[s][k], state[s][k]);   // (lane = address bits 7..2)
    }
    // 2c. field bytes -> 16-byte block -> ring (later steps copy from it) and memory, step after step
    const unsigned nsteps = (out_len + kStepBytes - 1u) / kStepBytes;
    // (a uniform branch around every step instead of a `break`: the compiler unrolls this form for the sixteen steps of
    // the 8-byte layouts too -- the rolled loop indexed the descriptors' registers through M0, two moves a field)
#pragma unroll
    for (unsigned s = 0; s < kMaxSteps; s++) {
        if (s < nsteps) {
        const unsigned opos = (64u * s + lane) * kBlock;
        const bool active = opos < out_len;
        unsigned out[kBlock / 4u];
        if (LAYOUT == 4u || LAYOUT == 6u) {
            // [2, 6 (, 4, 4)]: field 1 is 6 bytes at any byte address: three dwords
            unsigned lo[PERIOD], hi1 = 0;
#pragma unroll
            for (unsigned k = 0; k < PERIOD; k++) {
                // (v_alignbyte_b32 shifts by the low two bits of its third operand: the address itself)
                const unsigned a = (unsigned)state[s][k] >> 8, aw = a >> 2;
                const unsigned d0 = bufw[aw], d1 = bufw[aw + 1u];
                lo[k] = __builtin_amdgcn_alignbyte(d1, d0, a);
                if (k == 1) {
                    const unsigned d2 = bufw[aw + 2u];
                    hi1 = __builtin_amdgcn_alignbyte(d2, d1, a);
                }
            }
            out[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x05040100u);      // bytes 0, 1 of field 0, then bytes 0, 1 of field 1
            out[1] = (lo[1] >> 16) | (hi1 << 16);
            if (LAYOUT == 4u) {
                out[2] = lo[2 % PERIOD];
                out[3] = lo[3 % PERIOD];
            }
        } else {
            // [4, 4] and [4, 4, 4, 4]: every field is one dword at any byte address
#pragma unroll
            for (unsigned k = 0; k < PERIOD; k++) {
                const unsigned a = (unsigned)state[s][k] >> 8, aw = a >> 2;
                out[k] = __builtin_amdgcn_alignbyte(bufw[aw + 1u], bufw[aw], a);
            }
        }
        if (active) {
            if (kBlock == 16u) {
                const uint4 v = make_uint4(out[0], out[1], out[2 % (kBlock / 4u)], out[3 % (kBlock / 4u)]);
                *reinterpret_cast<uint4 *>(buf + opos) = v;
                if (dst_wide) {
                    gstore16(dst + opos, v);
                } else {
#pragma unroll 1
                    for (unsigned k = 0; k < 16u; k++)
                        dst[opos + k] = (uint8_t)(out[k >> 2] >> (8u * (k & 3u)));
                }
            } else {
                const uint2 v = make_uint2(out[0], out[1]);
                *reinterpret_cast<uint2 *>(buf + opos) = v;
                if (dst_wide) {
                    gstore8(dst + opos, v);
                } else {
#pragma unroll 1
                    for (unsigned k = 0; k < 8u; k++)
                        dst[opos + k] = (uint8_t)(out[k >> 2] >> (8u * (k & 3u)));
                }
            }
        }
        __syncthreads();
        }
    }
}

// One launch for every field-stream unit of a batch, whatever its layout (the two textures of a Hap Q Alpha frame are
// units of one array): the layout is uniform per wave.
__global__ __launch_bounds__(64) void snappy_decode_fields_kernel(const HapGpuDecodeUnit *__restrict__ units,
                                                                  unsigned unit_count, HapGpuDecodeJob *jobs)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[kBufBytes];
    __shared__ __attribute__((aligned(8))) uint2 masks[kHalves];
    __shared__ uint32_t coffs[kHalves + 2u];                       // ordinal of the half-tile's first element
    const unsigned lane = threadIdx.x;
    if (blockIdx.x >= unit_count)
        return;
    const HapGpuDecodeUnit u = units[blockIdx.x];
#ifdef SDF_ONLY
    // (instruction-count studies, tools/isa_by_line.py: one layout's code alone)
    if (u.kind != 0u)
        decode_fields_unit<SDF_ONLY>(u, jobs, buf, masks, coffs, lane);
    return;
#endif
    if (u.kind == HAPGPU_UNIT_SNAPPY_FIELDS4)
        decode_fields_unit<4u>(u, jobs, buf, masks, coffs, lane);
    else if (u.kind == HAPGPU_UNIT_SNAPPY_FIELDS2)
        decode_fields_unit<2u>(u, jobs, buf, masks, coffs, lane);
    else if (u.kind == HAPGPU_UNIT_SNAPPY_FIELDS26)
        decode_fields_unit<6u>(u, jobs, buf, masks, coffs, lane);
    else if (u.kind == HAPGPU_UNIT_SNAPPY_FIELDS44)
        decode_fields_unit<8u>(u, jobs, buf, masks, coffs, lane);
}


// ---- group tables for field streams that come without one ------------------------------------------------------------
// Frames written with HAPGPU_ENCODE_FINE_CHUNKS carry no private section: every 8 KiB fragment is a chunk of its own in
// the tables every Hap parser reads, so its boundaries are known -- but not the 64 places inside it from which the
// kernel above starts its lanes.  This kernel finds them: ONE LANE per fragment walks the fragment's tags (twice: count
// the elements, then note where every G-th begins) and writes the table of version 4 into scratch; the fragment's unit
// becomes a FIELDS unit that points there.  A lane's walk is ~450 dependent loads from memory -- 64 different cache lines
// per wave-instruction, which the memory system serves at about 100 G requests a second whatever the occupancy: 2.1 ms
// for the 243 000 fragments of 60 8K frames, three times the decode itself (the host takes this road from a few thousand
// fragments on; fewer are decoded by the generic kernel, one wavefront per fragment).  A cooperative form -- the wave
// stages its 64 fragments through LDS in coalesced 256-byte slices and the lanes walk out of LDS -- was built in round 5
// and is NO faster (2.2 ms): every slice is a load / barrier / walk / barrier round trip, and a wave runs as long as the
// busiest of its 64 fragments has elements.  Nor is a private 64-byte window per lane in LDS (four aligned 16-byte loads
// per refill, tags read from the slot, no barrier): some lane of the 64 needs its refill at almost every turn, so the
// wave waits for global loads as often as before -- 2.17 ms at 243 000 fragments, and 0.92 ms instead of 0.45 at 8 000,
// where nothing but a lane's own latency counts.  Cache hints on the tag loads do not help either: streaming (`nt`) loads
// make the pre-pass 1.9 ms SLOWER per 60 frames (the lines a lane comes back to are gone), loads past the L1 change nothing.
// This is the simplest of the three.
// What is not a field stream -- another encoder's chunk of the same size, an element off a field boundary -- stays a
// STREAM unit for the generic kernel: the walk checks the promises the table would have made.  (The kernel above checks
// them all again; its verdict, not this one's, is what protects memory.)
__device__ __forceinline__ unsigned load_tag32(gin_t src, unsigned cp, unsigned n)
{
    // bytes cp .. cp + 3 of the stream (zero beyond its end); unaligned dword loads are fine in global memory
    struct __attribute__((packed)) unaligned32 { uint32_t v; };
    if (cp + 4u <= n)
        return reinterpret_cast<const unaligned32 __attribute__((address_space(1))) *>(src + cp)->v;
    unsigned w = 0;
    for (unsigned k = 0; k < 4u && cp + k < n; k++)
        w |= (unsigned)src[cp + k] << (8u * k);
    return w;
}

// `work` == nullptr: the lane's unit is units[blockIdx.x * 64 + lane] -- a STREAM unit of a frame whose chunks are single
// fragments (length prefix + elements).  `work` != nullptr: the units the block scan listed ([0]: how many, then their
// indices) -- the 8 KiB pieces of a table-less stream of this library (plain hap.h frames), bare elements between two of
// the scan's marks.
__global__ __launch_bounds__(64) void guess_group_tables_kernel(HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                                                const uint32_t *work)
{
    unsigned idx = blockIdx.x * 64u + threadIdx.x;
    if (work) {
        if (idx >= work[0])
            return;
        idx = work[1u + idx];
    }
    if (idx >= unit_count)
        return;
    HapGpuDecodeUnit u = units[idx];
    const HapGpuDecodeJob *job = &jobs[u.job];
    if (!((job->reserved >> 16) & 1u) || job->group_tables == 0u || job->status != 0u)
        return;
    const unsigned layout = job->fields_period;
    const unsigned block = (layout == 4u || layout == 8u) ? 16u : 8u;
    // field starts inside a block, as a mask over its bytes: [2,6,4,4]: 0, 2, 8, 12; [4,4]: 0, 4; [2,6]: 0, 2; [4,4,4,4]: 0, 4, 8, 12
    const unsigned starts = layout == 4u ? 0x1105u : layout == 2u ? 0x11u : layout == 6u ? 0x05u : 0x1111u;
    unsigned hdr = 0, out_len = 0, n = 0;
    gin_t src;
    if (!work) {
        if (u.kind != HAPGPU_UNIT_SNAPPY_STREAM || u.aux != 0u || u.reserved != 0u)
            return;
        const gin_t base = (gin_t)u.src;
        // the stream's length prefix
        for (unsigned k = 0; k < 5u && k < u.src_len; k++) {
            const unsigned b = base[k];
            out_len |= (b & 0x7Fu) << (7u * k);
            if (!(b & 0x80u)) {
                hdr = k + 1u;
                break;
            }
        }
        if (hdr == 0u || out_len != u.dst_len)
            return;
        src = base + hdr;
        n = u.src_len - hdr;
    } else {
        // (what snappy_decode_fragment_kernel does with such a unit: the piece lies between marks b and b + 1 of its stream)
        if (u.kind != HAPGPU_UNIT_SNAPPY_BLOCK || !(u.reserved & HAPGPU_BLOCK_FINE))
            return;
        const HapGpuScanChunk *scan = (const HapGpuScanChunk *)u.aux;
        const unsigned b = (unsigned)u.reserved, marks = scan->expected_fine;
        if (!scan->ok || marks == 0u || scan->found_fine != marks || b + 1u > marks)
            return;
        const uint32_t *bpos = (const uint32_t *)scan->bpos;
        const unsigned from = bpos[b], to = bpos[b + 1u];
        if (from > to || to > bpos[marks])
            return;
        src = (gin_t)u.src + from;
        n = to - from;
        out_len = u.dst_len;
    }
    if (out_len == 0u || out_len > kFragBytes || (out_len % block) != 0u || n > kMaxFragCompressed)
        return;
    // first walk: count the elements, check what the table promises
    unsigned cp = 0, p = 0, count = 0;
    bool ok = true;
    while (cp < n && ok) {
        const unsigned w = load_tag32(src, cp, n);
        const unsigned kind = w & 3u, up = (w >> 2) & 63u;
        unsigned len, adv;
        if (kind == 0u) {
            ok = up <= 60u;
            len = (up == 60u ? ((w >> 8) & 255u) : up) + 1u;
            adv = len + (up == 60u ? 2u : 1u);
        } else {
            len = kind == 1u ? ((w >> 2) & 7u) + 4u : up + 1u;
            adv = kind + 1u;
            const unsigned off = kind == 1u ? (((w >> 5) & 7u) << 8) | ((w >> 8) & 255u) : (w >> 8) & 0xFFFFu;
            ok = kind != 3u && off >= block && (off % block) == 0u && off <= p;
        }
        ok = ok && ((starts >> (p % block)) & 1u) != 0u && (p & (kHalf - 1u)) + len <= kHalf;
        p += len;
        cp += adv;
        count += 1u;
    }
    // (an element that ends off a field boundary shows as the next one's start, or as the total)
    if (!ok || cp != n || p != out_len || 4u * count > out_len)
        return;
    // second walk: where every G-th element begins
    const unsigned G = (count + 63u) >> 6;
    gout_t table = (gout_t)(job->group_tables + (uint64_t)idx * HAP_GROUP_TABLE_BYTES);
    // (the table's 3-byte entries leave as dwords: the arena and 196 are multiples of 4)
    uint32_t __attribute__((address_space(1))) *table32 = (uint32_t __attribute__((address_space(1))) *)table;
    unsigned long long pending = 0;
    unsigned pending_bytes = 0, words = 0;
    unsigned g = 0, left = G, cp0 = 0, p0 = 0;
    cp = 0;
    p = 0;
    bool fits = true;
    while (cp < n) {
        const unsigned w = load_tag32(src, cp, n);
        const unsigned kind = w & 3u, up = (w >> 2) & 63u;
        const unsigned len = kind == 0u ? (up == 60u ? ((w >> 8) & 255u) : up) + 1u : kind == 1u ? ((w >> 2) & 7u) + 4u : up + 1u;
        const unsigned adv = kind == 0u ? len + (up == 60u ? 2u : 1u) : kind + 1u;
        p += len;
        cp += adv;
        if (--left == 0u || cp >= n) {
            const unsigned cs = cp - cp0, os = p - p0;
            fits = fits && cs < 4096u && os < 4096u;
            const unsigned entry = (cs | (os << 12)) & 0xFFFFFFu;
            pending |= (unsigned long long)entry << (8u * pending_bytes);
            pending_bytes += 3u;
            if (pending_bytes >= 4u) {
                table32[words++] = (uint32_t)pending;
                pending >>= 32;
                pending_bytes -= 4u;
            }
            g += 1u;
            left = G;
            cp0 = cp;
            p0 = p;
        }
    }
    if (pending_bytes != 0u || g < 64u) {
        table32[words++] = (uint32_t)pending;           // (the entries of the groups that do not exist are zero)
        for (; words < 48u; words++)
            table32[words] = 0u;
    }
    table32[48] = count;                                // bytes 192, 193: the element count; 194, 195: zero
    if (!fits)
        return;
    // the unit becomes a field-stream fragment: bare elements, its table, the readable bytes behind it
    const uint64_t end = (uint64_t)(uintptr_t)src + n, section_end = job->payload + job->payload_len;
    u.src = (uint64_t)(uintptr_t)src;
    u.src_len = n;
    u.kind = layout == 4u ? HAPGPU_UNIT_SNAPPY_FIELDS4 : layout == 2u ? HAPGPU_UNIT_SNAPPY_FIELDS2
           : layout == 8u ? HAPGPU_UNIT_SNAPPY_FIELDS44 : HAPGPU_UNIT_SNAPPY_FIELDS26;
    u.aux = (uint64_t)(uintptr_t)table;
    u.reserved = section_end > end ? (section_end - end < 15u ? section_end - end : 15u) : 0u;
    units[idx] = u;
}

} // namespace

// work == nullptr: every unit of the call is looked at (frames whose chunks are single fragments); else the `work_slots`
// units the block scan may have listed in `work` (the count is on the device)
extern "C" int hapgpu_launch_guess_group_tables(HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                                const uint32_t *work, unsigned work_slots, hipStream_t stream)
{
    const unsigned lanes = work ? work_slots : unit_count;
    if (unit_count == 0 || lanes == 0)
        return 0;
    // Every lane reads its own stream, a few bytes per turn: a wavefront's 64 lanes keep 64 cache lines alive, and with
    // sixteen wavefronts on a CU none of them survives in its 32 KiB L1 until the lane's next element (every turn then
    // comes from the L2).  Dynamic LDS the kernel never touches keeps the wavefronts per CU down to what the L1 holds.
    static int lds_bytes = -1;
    if (lds_bytes < 0) {
        const char *e = HAP_AB_ENV("HAP_AMD_GUESS_LDS");
        lds_bytes = e ? atoi(e) : kGuessLdsBytes;
        if (lds_bytes > 65536)
            lds_bytes = 65536;
    }
    hipLaunchKernelGGL(guess_group_tables_kernel, dim3((lanes + 63u) / 64u), dim3(64), (unsigned)lds_bytes, stream, units, unit_count, jobs, work);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// fields_kinds: bit 0 = [2, 6, 4, 4] units present (DXT5 / YCoCg-DXT5), bit 1 = [4, 4] units (DXT1), bit 2 = [2, 6] (RGTC1),
// bit 3 = [4, 4, 4, 4] (opaque 16-byte blocks)
extern "C" int hapgpu_launch_snappy_decode_fields(const HapGpuDecodeUnit *units, unsigned unit_count, HapGpuDecodeJob *jobs,
                                                  unsigned fields_kinds, hipStream_t stream)
{
    if (unit_count == 0)
        return 0;
    if (fields_kinds & 15u)
        hipLaunchKernelGGL(snappy_decode_fields_kernel, dim3(unit_count), dim3(64), SDF_DYN_LDS, stream, units, unit_count, jobs);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
