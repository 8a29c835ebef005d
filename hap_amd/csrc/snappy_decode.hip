// snappy_decode.hip -- decode planner + the generic per-unit Snappy decoder for gfx950.
//
// Replaces, for HapDecode, the reference's planning loop (hap.c:794-838: per-chunk
// snappy_uncompressed_length + running output offsets), its HapDecodeCallback fan-out
// (hap.c:852-862), and the worker hap_decode_chunk (hap.c:606-642: snappy_uncompress | memcpy).
//
//   decode_plan_kernel    one wavefront per texture: reads every chunk's varint, prefix-sums the
//                         output offsets, applies the reference's error precedence, and emits the
//                         units of chunks without a fragment table (whole stream, raw copy).
//   decode_expand_kernel  one wavefront per chunk that has fragment-table entries: one unit per
//                         fragment (prefix sum over the entries).
//   snappy_decode_fragment_kernel
//                         the generic decoder, one wavefront per unit, any valid Snappy stream:
//                         a 64-byte window of compressed bytes is parsed speculatively at every byte
//                         position, the real element chain is found by pointer doubling, and the
//                         window's output is produced 64 units (bytes, or 2 / 4 bytes when the table
//                         promises such granularity) per step through an owner map; the last RING
//                         bytes of output live in LDS so that back-references are LDS -> LDS.
//                         Used for frames from other encoders (one unit per 64 KiB block found by the
//                         scan below with a 2 KiB ring, or one unit per chunk with a 32 KiB ring; older
//                         bytes are re-read from memory), for fragment tables of version 1, and as the
//                         fallback whenever a table turns out not to describe its streams.
//   scan_walk_kernel / scan_merge_kernel / scan_find_kernel / scan_decide_kernel
//                         the block scan: where in the compressed bytes of a stream without a table each
//                         64 KiB block of output begins (libsnappy's blocks are independent) -- or each
//                         8 KiB (the fragments of this library's own table-less frames) --, found in
//                         parallel over 4 KiB segments of compressed bytes without producing output: see
//                         the comment in front of them.
//
// Frames written by this library with the version-3 table ("field streams": DXT5, YCoCg-DXT5, DXT1,
// large RGTC1 planes) are decoded by the block-per-lane kernel of snappy_decode_fields.hip instead.
// HBM traffic: compressed bytes read once + output written once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "hapgpu_abi.h"
#include "measurement_guard.h"

namespace {

constexpr unsigned kResBadFrame = 3, kResTooSmall = 2, kResInternal = 4;
constexpr unsigned kCopyPiece = 65536;   // raw chunks are copied in pieces of this size

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------

// varint32 length prefix (<= 5 bytes, fifth byte < 16). Returns header bytes, 0 if malformed.
__device__ unsigned parse_varint(const uint8_t *p, unsigned avail, unsigned *value)
{
    unsigned v = 0;
    for (unsigned i = 0; i < 5; i++) {
        if (i >= avail)
            return 0;
        const unsigned b = p[i];
        if (i == 4 && b >= 16)
            return 0;
        v |= (b & 0x7fu) << (7 * i);
        if (!(b & 0x80u)) {
            *value = v;
            return i + 1;
        }
    }
    return 0;
}

__device__ void emit_copy_units(HapGpuDecodeUnit *u, unsigned slots, const uint8_t *src, uint8_t *dst,
                                unsigned len, unsigned job)
{
    for (unsigned k = 0; k < slots; k++) {
        const unsigned long long at = (unsigned long long)k * kCopyPiece;
        HapGpuDecodeUnit w;
        w.src = (uint64_t)(src + at);
        w.dst = (uint64_t)(dst + at);
        w.src_len = at < len ? min(kCopyPiece, (unsigned)(len - at)) : 0u;
        w.dst_len = w.src_len;
        w.kind = w.src_len ? HAPGPU_UNIT_COPY : HAPGPU_UNIT_SKIP;
        w.job = job;
        w.aux = 0;
        w.reserved = 0;
        u[k] = w;
    }
}

__global__ __launch_bounds__(64) void decode_plan_kernel(HapGpuDecodeJob *jobs, unsigned job_count)
{
    const unsigned j = blockIdx.x, lane = threadIdx.x;
    if (j >= job_count)
        return;
    HapGpuDecodeJob *job = &jobs[j];
    HapGpuDecodeUnit *units = (HapGpuDecodeUnit *)job->units;
    const uint8_t *payload = (const uint8_t *)job->payload;
    uint8_t *dst = (uint8_t *)job->dst;

    if (job->mode != HAPGPU_JOB_COMPLEX) {
        if (lane != 0)
            return;
        unsigned status = 0;
        unsigned long long used = 0;
        // (every unit slot reads SKIP already: hapgpu_k_decode_plan clears the array before this kernel)
        if (job->mode == HAPGPU_JOB_RAW) {                 // reference hap.c:905-916
            used = job->payload_len;
            if (used > job->dst_cap)
                status = kResTooSmall;
            else
                emit_copy_units(units, job->unit_count, payload, dst, (unsigned)used, j);
        } else {                                           // reference hap.c:885-904
            unsigned n = 0;
            const unsigned h = parse_varint(payload, (unsigned)min(job->payload_len, (uint64_t)5), &n);
            if (!h)
                status = kResInternal;
            else if (n > job->dst_cap)
                status = kResTooSmall;
            else {
                HapGpuDecodeUnit w;
                w.src = (uint64_t)payload; w.dst = (uint64_t)dst;
                w.src_len = (unsigned)job->payload_len; w.dst_len = n;
                w.kind = HAPGPU_UNIT_SNAPPY_STREAM; w.job = j;
                w.aux = job->unit_count - 1u; w.reserved = 0;       // slots for the block scan
                units[0] = w;
                used = n;
            }
        }
        job->bytes_used = used;
        job->status = status;
        return;
    }

    // complex: reference hap.c:794-843
    HapGpuChunkIn *chunks = (HapGpuChunkIn *)job->chunks;
    const unsigned n = job->chunk_count;
    const uint32_t *frag_sizes = (const uint32_t *)job->frag_sizes;
    const unsigned frag_bytes = 1u << job->frag_log2;
    const unsigned per_chunk = (frag_sizes && n) ? job->frag_entries / n : 0u;
    unsigned long long run = 0;
    bool bad_varint = false, bad_codec = false, index_bad = false;
    for (unsigned base = 0; base < n; base += 64) {
        const unsigned i = base + lane;
        unsigned out_len = 0, hdr = 0, codec = 0;
        bool bad = false, wanted = false;
        HapGpuChunkIn c = {};
        if (i < n) {
            c = chunks[i];
            codec = c.codec & 0xFFu;
            wanted = (c.codec >> 31) == 0;
            if (codec == HAP_NIBBLE_SNAPPY) {
                hdr = parse_varint(payload + c.src_off, c.src_len, &out_len);
                bad = hdr == 0;
            } else {
                out_len = c.src_len;
                if (codec != HAP_NIBBLE_NONE && wanted)
                    bad_codec = true;
            }
        }
        // the reference stops at the first chunk whose length prefix is malformed
        const unsigned long long badmask = __ballot(bad);
        if (badmask) {
            bad_varint = true;
            break;
        }
        // inclusive scan of out_len across the wave
        unsigned long long incl = out_len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(incl, d);
            if ((int)lane >= d)
                incl += up;
        }
        const unsigned long long my_off = run + incl - out_len;
        run += __shfl(incl, 63);
        if (i < n) {
            HapGpuDecodeUnit *u = units + c.unit_first;                 // (all SKIP so far: cleared before the launch)
            if (!wanted || my_off + out_len > job->dst_cap) {
                // not requested by the client / will be rejected as Buffer_Too_Small below
            } else if (codec == HAP_NIBBLE_NONE) {
                emit_copy_units(u, c.unit_count, payload + c.src_off, dst + my_off, c.src_len, j);
            } else if (codec == HAP_NIBBLE_SNAPPY) {
                // chunks with a fragment table are expanded into units by decode_expand_kernel (one wavefront per
                // chunk); here only the chunk-level facts are recorded
                const bool indexed = per_chunk && c.unit_count == per_chunk &&
                                     (unsigned long long)per_chunk * frag_bytes >= out_len &&
                                     (unsigned long long)(per_chunk - 1) * frag_bytes < out_len;
                if (indexed) {
                    chunks[i].plan_expand = 1u;
                    chunks[i].plan_hdr = hdr;
                    chunks[i].plan_out_len = out_len;
                    chunks[i].plan_out_off = my_off;
                }
                if (!indexed) {
                    // a table is present but does not describe this chunk: the host redoes the whole
                    // texture without it (the launch may not even include the whole-stream kernel)
                    if (per_chunk)
                        index_bad = true;
                    HapGpuDecodeUnit w;
                    w.src = (uint64_t)(payload + c.src_off);
                    w.dst = (uint64_t)(dst + my_off);
                    w.src_len = c.src_len;
                    w.dst_len = out_len;
                    w.kind = HAPGPU_UNIT_SNAPPY_STREAM;
                    w.job = j;
                    w.aux = c.unit_count - 1u;                    // slots for the block scan
                    w.reserved = 0;
                    u[0] = w;
                }
            }
        }
    }
    bad_codec = __ballot(bad_codec) != 0;
    index_bad = __ballot(index_bad) != 0;
    if (lane == 0) {
        unsigned status = 0;
        if (bad_varint)
            status = kResBadFrame;
        else if (run > job->dst_cap)
            status = kResTooSmall;
        else if (bad_codec)
            status = kResBadFrame;             // reference hap.c:637-640 via 867-875
        else if (index_bad)
            status = HAPGPU_STATUS_INDEX_MISMATCH;
        job->bytes_used = run;
        job->status = status;
    }
}

// One wavefront per chunk that carries fragment-table entries: compressed offsets by a prefix sum over the entries,
// one unit per fragment.  (The planner above walks chunks; a 16K texture has 512 fragments per chunk.)
__global__ __launch_bounds__(64) void decode_expand_kernel(HapGpuDecodeJob *jobs, unsigned job_count)
{
    const unsigned j = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
    if (j >= job_count)
        return;
    HapGpuDecodeJob *job = &jobs[j];
    if (job->mode != HAPGPU_JOB_COMPLEX || i >= job->chunk_count)
        return;
    const HapGpuChunkIn c = ((const HapGpuChunkIn *)job->chunks)[i];
    if (!c.plan_expand)
        return;
    const unsigned n = job->chunk_count;
    const uint32_t *frag_sizes = (const uint32_t *)job->frag_sizes;
    const unsigned frag_bytes = 1u << job->frag_log2;
    const unsigned per_chunk = job->frag_entries / n;
    const uint32_t *fs = frag_sizes + c.frag_first;
    const uint8_t *payload = (const uint8_t *)job->payload;
    uint8_t *dst = (uint8_t *)job->dst;
    HapGpuDecodeUnit *u = (HapGpuDecodeUnit *)job->units + c.unit_first;
    const unsigned gran = job->reserved & 0xFFu, window256 = (job->reserved >> 8) & 0xFFu;
    unsigned kind = gran == 2 ? HAPGPU_UNIT_SNAPPY_FRAGMENT32 : gran == 1 ? HAPGPU_UNIT_SNAPPY_FRAGMENT16 : HAPGPU_UNIT_SNAPPY_FRAGMENT;
    if (job->frag_log2 == 13u && window256 != 0 && window256 <= HAP_FRAGMENT_WINDOW_256)
        kind |= HAPGPU_UNIT_WINDOWED;
    const bool fields = job->fields_period != 0u && job->group_tables != 0u;
    if (fields)       // field stream (table version 3): block-per-lane decoder, with the fragment's group table
        kind = job->fields_period == 4u ? HAPGPU_UNIT_SNAPPY_FIELDS4
             : job->fields_period == 2u ? HAPGPU_UNIT_SNAPPY_FIELDS2
             : job->fields_period == 8u ? HAPGPU_UNIT_SNAPPY_FIELDS44 : HAPGPU_UNIT_SNAPPY_FIELDS26;
    unsigned long long run = c.plan_hdr;
    for (unsigned base = 0; base < per_chunk; base += 64u) {
        const unsigned k = base + lane;
        const unsigned sz = k < per_chunk ? fs[k] : 0u;
        unsigned long long incl = sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(incl, d);
            if ((int)lane >= d)
                incl += up;
        }
        const unsigned long long at = run + incl - sz;
        run += __shfl(incl, 63);
        if (k < per_chunk) {
            HapGpuDecodeUnit w;
            w.src = (uint64_t)(payload + c.src_off + at);
            w.dst = (uint64_t)(dst + c.plan_out_off + (unsigned long long)k * frag_bytes);
            w.src_len = sz;
            w.dst_len = min(frag_bytes, c.plan_out_len - k * frag_bytes);
            w.kind = kind;
            w.job = j;
            w.aux = fields ? job->group_tables + (uint64_t)(c.frag_first + k) * HAP_GROUP_TABLE_BYTES : 0u;
            // bytes of the texture section that follow the fragment (up to 15): the decoder may fetch its last
            // 16-byte piece whole when they exist
            const uint64_t end = (uint64_t)c.src_off + at + sz;
            const uint64_t after = job->payload_len > end ? job->payload_len - end : 0u;
            w.reserved = after < 15u ? after : 15u;
            // a table whose entries run past the chunk is caught below; keep the unit harmless until then
            if (end > (uint64_t)c.src_off + c.src_len)
                w.kind = HAPGPU_UNIT_SKIP;
            u[k] = w;
        }
    }
    // the entries must add up to the chunk: otherwise the host redoes the whole texture without the table
    if (lane == 0 && run != c.src_len)
        atomicCAS(&job->status, 0u, HAPGPU_STATUS_INDEX_MISMATCH);
}

// ------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------

constexpr unsigned kSegment = 4096;

__device__ __forceinline__ unsigned uniform(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

// Cooperative byte-range copy global->global for raw chunks.
__device__ void wave_copy(uint8_t *dst, const uint8_t *src, unsigned len, unsigned lane)
{
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        const unsigned wide = len >> 4;
        for (unsigned i = lane; i < wide; i += 64)
            reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        for (unsigned i = (wide << 4) + lane; i < len; i += 64)
            dst[i] = src[i];
    } else if ((((uintptr_t)dst | (uintptr_t)src) & 3u) == 0) {
        const unsigned words = len >> 2;
        for (unsigned i = lane; i < words; i += 64)
            reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
        for (unsigned i = (words << 2) + lane; i < len; i += 64)
            dst[i] = src[i];
    } else {
        for (unsigned i = lane; i < len; i += 64)
            dst[i] = src[i];
    }
}

// Writes ring positions [from, to) to global memory; `to - from` <= RING.
template <unsigned RING>
__device__ void flush_ring(const uint8_t *ring, uint8_t *dst, unsigned from, unsigned to, unsigned lane)
{
    // head up to 16-byte alignment of the global address, then 16 B per lane, then tail
    unsigned at = from;
    const unsigned mis = (unsigned)((uintptr_t)(dst + at) & 15u);
    if (mis) {
        const unsigned head = min(16u - mis, to - at);
        if (lane < head)
            dst[at + lane] = ring[(at + lane) & (RING - 1)];
        at += head;
    }
    const unsigned wide = (to - at) >> 4;
    if (((at & 15u) == 0)) {
        for (unsigned i = lane; i < wide; i += 64) {
            const unsigned x = at + (i << 4);
            *reinterpret_cast<uint4 *>(dst + x) = *reinterpret_cast<const uint4 *>(ring + (x & (RING - 1)));
        }
    } else {
        // ring offset not 16-aligned relative to the global address: assemble from bytes
        for (unsigned i = lane; i < wide; i += 64) {
            const unsigned x = at + (i << 4);
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned y = x + 4 * k;
                w[k] = (unsigned)ring[y & (RING - 1)] | ((unsigned)ring[(y + 1) & (RING - 1)] << 8) |
                       ((unsigned)ring[(y + 2) & (RING - 1)] << 16) | ((unsigned)ring[(y + 3) & (RING - 1)] << 24);
            }
            *reinterpret_cast<uint4 *>(dst + x) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    at += wide << 4;
    if (at + lane < to)
        dst[at + lane] = ring[(at + lane) & (RING - 1)];
}


// ------------------------------------------------------------------------------------------
// window-parallel fragment decoder
// ------------------------------------------------------------------------------------------
//
// Walking a stream element by element costs ~40 scalar instructions and several dependent LDS round
// trips per element.  This kernel works on a 64-byte window of compressed bytes at a time:
//
//   1. every lane parses "the element that would start at my byte" (speculatively, in parallel);
//   2. which of those are real is a chain from lane 0: two rounds of pointer doubling with
//      ds_bpermute give each lane the set of its next 4 chain members, and the scalar unit then
//      hops 4 elements per iteration;
//   3. a DPP prefix sum over the real elements gives every element its output position;
//   4. the window's output is produced 64 bytes per step, one byte per lane: lanes find their
//      element through an owner map in LDS, compute their source (literal byte in the staging
//      window, or an earlier output byte in the LDS ring) and sources that fall inside the same
//      64-byte step are chased with pointer jumping in registers (<= 6 rounds) before one LDS
//      gather + one LDS store.
//
// Failure semantics of snappy_uncompress: any malformed element, bad offset or length mismatch fails
// the unit (hap.c:617-628: Bad_Frame / Internal_Error; a fragment unit: table mismatch -> fallback).

#ifndef HAP_V2_IN_BYTES
#define HAP_V2_IN_BYTES 1024
#endif
#ifndef HAP_V2_OWNER_BYTES
#define HAP_V2_OWNER_BYTES 512
#endif
constexpr unsigned kV2InBytes = HAP_V2_IN_BYTES, kV2InGranule = kV2InBytes / 2u;   // staged input: two granules
constexpr unsigned kOwnerBytes = HAP_V2_OWNER_BYTES;      // output bytes handled per window pass
constexpr unsigned kFragmentTail = kV2InBytes + kOwnerBytes + 64u;   // LDS of the v2 kernel beyond its ring

__device__ __forceinline__ int dpp_row_shr(int identity, int v, int n)
{
    switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xF, 0xF, false);
    case 4: return __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xF, 0xF, false);
    }
}

// wave64 inclusive scans with DPP (row shifts inside each 16-lane row, then row broadcasts)
__device__ __forceinline__ int wave_scan_add(int v)
{
    v += dpp_row_shr(0, v, 1);
    v += dpp_row_shr(0, v, 2);
    v += dpp_row_shr(0, v, 4);
    v += dpp_row_shr(0, v, 8);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ int wave_scan_max(int v)
{
    v = max(v, dpp_row_shr(0, v, 1));
    v = max(v, dpp_row_shr(0, v, 2));
    v = max(v, dpp_row_shr(0, v, 4));
    v = max(v, dpp_row_shr(0, v, 8));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));
    return v;
}

__device__ __forceinline__ unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

__device__ __forceinline__ int lane_gather(int v, unsigned src_lane)
{
    return __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), v);
}

// Lanes on the element chain that starts at lane `from` of a 64-byte window (0: the element at `from` is a stopper).
// nxt: lane of the element after this lane's (64: beyond the window or none).  Pointer doubling gives every lane the
// set of the next 2^kChainRounds chain elements from it; the scalar unit then hops along the chain that many at a time.
#ifndef HAP_CHAIN_ROUNDS
#define HAP_CHAIN_ROUNDS 3
#endif
__device__ __forceinline__ unsigned long long window_chain(bool stopper, unsigned nxt, unsigned lane, unsigned from)
{
    unsigned long long mask = stopper ? 0ull : (1ull << lane);
    unsigned j = nxt;
#pragma unroll
    for (int round = 0; round < HAP_CHAIN_ROUNDS; round++) {
        const unsigned g_j = (unsigned)lane_gather((int)j, j & 63u);
        const unsigned g_lo = (unsigned)lane_gather((int)(unsigned)mask, j & 63u);
        const unsigned g_hi = (unsigned)lane_gather((int)(unsigned)(mask >> 32), j & 63u);
        mask |= j < 64u ? (((unsigned long long)g_hi << 32) | g_lo) : 0ull;
        j = j < 64u ? g_j : 64u;
    }
    unsigned long long T = 0;
    for (unsigned s = from; s < 64u;) {
        const unsigned m_lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mask, (int)s);
        const unsigned m_hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mask >> 32), (int)s);
        T |= ((unsigned long long)m_hi << 32) | m_lo;
        s = (unsigned)__builtin_amdgcn_readlane((int)j, (int)s);
    }
    return T;
}

// GRAN = 2: every element of the unit has even length and (copies) even offset, so one lane moves
// 16 bits (FRAGMENT16 units, produced by hap_amd's 16-bit granular compressor).
// SLIDE: the unit is a fragment larger than the ring whose copies all stay within RING - 1 KiB (promised by the
// fragment table, checked here): finished output leaves the ring in 1 KiB segments while decoding goes on.
// phase (STREAM launches): 0 every unit; 1 the 8 KiB BLOCK units the block scan listed in `work` ([0]: how many, then
// their indices in `units`); 2 everything else, after phase 1
template <unsigned RING, bool STREAM, unsigned GRAN, bool SLIDE = false>
__global__ __launch_bounds__(64) void snappy_decode_fragment_kernel(const HapGpuDecodeUnit *__restrict__ units,
                                                                    unsigned unit_count, HapGpuDecodeJob *jobs, unsigned phase,
                                                                    const uint32_t *__restrict__ work)
{
    // Statically sized LDS when it fits the 64 KiB static limit: the compiler then knows every LDS address
    // offset at compile time (with a dynamic array each address computation carries an extra add of the base).
    constexpr bool kStaticLds = RING + kFragmentTail <= 65536u;
    extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
    __shared__ __attribute__((aligned(16))) uint8_t static_lds[kStaticLds ? RING + kFragmentTail : 16u];
    uint8_t *const smem = kStaticLds ? static_lds : dynamic_lds;
    uint8_t *ring = smem;                                          // [0, RING): output
    uint32_t *inw = reinterpret_cast<uint32_t *>(smem + RING);     // [RING, RING+2048): staged input
    uint8_t *owner = smem + RING + kV2InBytes;                       // [.., +1024+64): owner map
    const unsigned lane = threadIdx.x;
    if (blockIdx.x >= unit_count)
        return;
    unsigned unit_index = blockIdx.x;
    if (STREAM && phase == 1u) {
        if (blockIdx.x >= work[0])
            return;
        unit_index = work[1u + blockIdx.x];
    }
    const HapGpuDecodeUnit u = units[unit_index];
    if (u.kind == HAPGPU_UNIT_SKIP)
        return;
    const bool fine_unit = STREAM && u.kind == HAPGPU_UNIT_SNAPPY_BLOCK && (u.reserved & HAPGPU_BLOCK_FINE) != 0ull;
    if (STREAM && fine_unit != (phase == 1u))
        return;
    HapGpuDecodeJob *job = &jobs[u.job];
    if (job->status != 0)
        return;
    if (u.kind == HAPGPU_UNIT_COPY) {
        if (STREAM)      // raw chunks ride with the stream launch
            wave_copy((uint8_t *)u.dst, (const uint8_t *)u.src, u.src_len, lane);
        return;
    }
    if (STREAM ? (u.kind != HAPGPU_UNIT_SNAPPY_STREAM && u.kind != HAPGPU_UNIT_SNAPPY_BLOCK)
               : u.kind != ((GRAN == 4 ? HAPGPU_UNIT_SNAPPY_FRAGMENT32
                                       : GRAN == 2 ? HAPGPU_UNIT_SNAPPY_FRAGMENT16 : HAPGPU_UNIT_SNAPPY_FRAGMENT) |
                            (SLIDE ? HAPGPU_UNIT_WINDOWED : 0u)))
        return;
    const uint8_t *src = (const uint8_t *)u.src;
    unsigned src_len = u.src_len;
    if (STREAM && (u.kind == HAPGPU_UNIT_SNAPPY_BLOCK || u.reserved != 0u)) {
        // a stream the block scan looked at: its BLOCK units run when every block start was found, else the stream unit
        const HapGpuScanChunk *scan = (const HapGpuScanChunk *)(u.kind == HAPGPU_UNIT_SNAPPY_BLOCK ? u.aux : u.reserved);
        // 2: every 8 KiB mark was found (a table-less stream of this library: its fragments) -- the fine BLOCK units run;
        // 1: every 64 KiB mark (libsnappy's blocks) -- the coarse ones; 0: the stream unit
        const bool fine_on = scan->expected_fine != 0u;
        // (fine_failed: written by the fine units of phase 1, read here by the others in phase 2)
        const bool fine_ok = fine_on && scan->found_fine == scan->expected_fine &&
                             (fine_unit || __hip_atomic_load(&scan->fine_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u);
        const unsigned mode = scan->ok == 0u ? 0u : fine_ok ? 2u : scan->found == scan->expected ? 1u : 0u;
        const unsigned mine = u.kind != HAPGPU_UNIT_SNAPPY_BLOCK ? 0u : (u.reserved & HAPGPU_BLOCK_FINE) ? 2u : 1u;
        if (mode != mine)
            return;
        if (mine != 0u) {
            const uint32_t *bpos = (const uint32_t *)scan->bpos;
            const unsigned b = (unsigned)u.reserved;
            const unsigned marks = fine_on ? scan->expected_fine : scan->expected;          // words of bpos, + 1 for the end
            const unsigned first = mine == 2u ? b : fine_on ? 8u * b : b;
            const unsigned last = mine == 2u ? b + 1u : fine_on ? min(8u * b + 8u, marks) : b + 1u;
            // (the block positions live in scratch the scan fills: whatever it left there, a block never reaches
            // outside its stream -- the stream is decoded by the second phase's units / the frame again without the scan)
            const bool in_range = first < marks && last <= marks;
            const unsigned from = in_range ? bpos[first] : 1u, to = in_range ? bpos[last] : 0u;
            if (!in_range || from > to || to > bpos[marks]) {      // (the last word is the stream's end, written with `ok`)
                if (lane == 0) {
                    if (mine == 2u && phase == 1u)
                        __hip_atomic_store(&((HapGpuScanChunk *)u.aux)->fine_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        atomicCAS(&jobs[u.job].status, 0u, HAPGPU_STATUS_INDEX_MISMATCH);
                }
                return;
            }
            src += from;
            src_len = to - from;
        }
    }
    constexpr bool kFlushEarly = STREAM || SLIDE;
    // rings below 16 KiB (windowed fragments; streams decoded with many waves per CU) give finished output back in
    // 1 KiB pieces
    constexpr unsigned kFlushSegment = (SLIDE || RING < 16384u) ? 1024u : kSegment;
    constexpr unsigned kFlushAt = (SLIDE || RING < 16384u) ? 1024u : 2u * kSegment;

    uint8_t *dst = (uint8_t *)u.dst;
    const unsigned shift = (unsigned)((uintptr_t)src & 15u);
    const uint8_t *src_al = src - shift;
    const unsigned in_end = shift + src_len;
    const unsigned out_len = u.dst_len;

    auto load_granule = [&](unsigned g) -> uint4 {
        const unsigned x = g * kV2InGranule + lane * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (lane * 16u >= kV2InGranule) {
            // (granules smaller than 64 x 16 bytes: the upper lanes have nothing to fetch)
        } else if (x >= shift && x + 16u <= in_end) {
            v = *reinterpret_cast<const uint4 *>(src_al + x);
        } else if (x + 16u > shift && x < in_end) {
            unsigned w[4] = {0, 0, 0, 0};
            for (unsigned k = 0; k < 16; k++) {
                const unsigned y = x + k;
                if (y >= shift && y < in_end)
                    w[k >> 2] |= (unsigned)src_al[y] << (8 * (k & 3));
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return v;
    };
    auto store_granule = [&](unsigned g, uint4 v) {
        if (lane * 16u < kV2InGranule)
            *reinterpret_cast<uint4 *>(smem + RING + ((g & 1u) * kV2InGranule) + lane * 16u) = v;
    };

    unsigned ip = shift;
    const unsigned granules = (in_end + kV2InGranule - 1) / kV2InGranule;
    store_granule(0, load_granule(0));
    store_granule(1, load_granule(1));
    unsigned next_g = 2, in_hi = 2 * kV2InGranule;
    uint4 pend = make_uint4(0, 0, 0, 0);
    bool pend_valid = false;
    __syncthreads();

    bool failed = !kFlushEarly && out_len > RING;
    unsigned op = 0, flushed = 0;
    if (STREAM && u.kind == HAPGPU_UNIT_SNAPPY_STREAM) {
        // skip the length prefix (validated by the plan kernel); BLOCK units are bare elements
        unsigned b;
        do {
            b = smem[RING + (ip & (kV2InBytes - 1))];
            ip++;
        } while ((b & 0x80u) && ip < in_end);
        ip = uniform(ip);
    }

    while (!failed && ip < in_end) {
        // ---- staging window: keep >= 384 bytes ahead of ip (a window's last element may be a 256-byte literal) ----
        if (!pend_valid && next_g < granules && ip + 3 * (kV2InGranule / 2) >= in_hi) {
            pend = load_granule(next_g);
            pend_valid = true;
        }
        if (ip + 384u > in_hi && in_hi < granules * kV2InGranule) {
            const unsigned g = ip / kV2InGranule;
            if (g + 1 == next_g) {
                store_granule(next_g, pend_valid ? pend : load_granule(next_g));
                next_g += 1;
            } else if (g >= next_g) {
                store_granule(g, load_granule(g));
                store_granule(g + 1, load_granule(g + 1));
                next_g = g + 2;
            }
            pend_valid = false;
            in_hi = next_g * kV2InGranule;
            __syncthreads();
        }

        // ---- 1. speculative parse: the element that would start at coordinate ip + lane ----
        const unsigned x = ip + lane;
        const unsigned wi = x >> 2, sh = x & 3u;
        const unsigned w0 = inw[wi & (kV2InBytes / 4u - 1u)], w1 = inw[(wi + 1) & (kV2InBytes / 4u - 1u)];
        const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, sh);       // bytes x .. x+3
        const unsigned hi = (w1 >> (8u * sh)) & 0xFFu;                     // byte x+4 (copy-4 / 4 length bytes)
        const unsigned tag = lo & 0xFFu, kind = tag & 3u;
        unsigned len, off = 0, hdr;
        bool special = false;                   // literal with a 2..4 byte length: taken alone by the slow path
        if (kind == 0) {
            len = (tag >> 2) + 1u;
            hdr = 1;
            if (len == 61u) {                   // one length byte: up to 256 bytes, handled like any element
                len = ((lo >> 8) & 0xFFu) + 1u;
                hdr = 2;
            } else if (len > 61u) {
                special = true;
                hdr = 1u + (len - 60u);
            }
        } else if (kind == 1) {
            len = 4u + ((tag >> 2) & 7u);
            off = ((tag >> 5) << 8) | ((lo >> 8) & 0xFFu);
            hdr = 2;
        } else if (kind == 2) {
            len = (tag >> 2) + 1u;
            off = (lo >> 8) & 0xFFFFu;
            hdr = 3;
        } else {
            len = (tag >> 2) + 1u;
            off = (lo >> 8) | (hi << 24);
            hdr = 5;
        }
        const unsigned tokbytes = hdr + (kind == 0 ? len : 0u);
        // an element that does not fit the input, or a long literal, ends the window before it
        const bool stopper = special || x >= in_end || tokbytes > in_end - x;
        const unsigned nxt = stopper ? 64u : min(lane + tokbytes, 64u);

        // ---- 2. chain membership from lane 0 ----
        unsigned long long T = window_chain(stopper, nxt, lane, 0u);

        if (T == 0) {
            // the element at ip is a long literal (or malformed): serial path, one element
            const unsigned lo0 = uniform(lo), hi0 = uniform(hi);
            const unsigned tag0 = lo0 & 0xFFu;
            if ((tag0 & 3u) != 0 || (tag0 >> 2) < 60u) { failed = true; break; }
            const unsigned extra = (tag0 >> 2) - 59u;
            const unsigned long long field = (((unsigned long long)hi0 << 32) | lo0) >> 8;
            const unsigned v = (unsigned)(extra == 4 ? field : (field & ((1ull << (8 * extra)) - 1ull)));
            const unsigned h = 1u + extra;
            if (h > in_end - ip || v == 0xFFFFFFFFu) { failed = true; break; }
            const unsigned llen = v + 1u;
            if (llen > in_end - ip - h || llen > out_len - op || (llen & (GRAN - 1u))) { failed = true; break; }
            ip += h;
            for (unsigned done = 0; done < llen; done += 64u) {
                const unsigned n = min(64u, llen - done);
                const bool staged = ip + done + n <= in_hi;
                if (lane < n) {
                    const unsigned y = ip + done + lane;
                    ring[(op + lane) & (RING - 1)] = staged ? smem[RING + (y & (kV2InBytes - 1))] : src_al[y];
                }
                op += n;
                if (kFlushEarly && op - flushed >= kFlushAt) {
                    const unsigned upto = op & ~(kFlushSegment - 1);
                    flush_ring<RING>(ring, dst, flushed, upto, lane);
                    flushed = upto;
                }
            }
            ip += llen;
            continue;
        }

        // ---- 3. output positions ----
        bool is_tok = __builtin_amdgcn_inverse_ballot_w64(T);
        int incl = wave_scan_add(is_tok ? (int)len : 0);
        unsigned o_t = (unsigned)incl - (is_tok ? len : 0u);
        // keep at most kOwnerBytes of output per pass (prefix-closed because o_t is monotone)
        T &= ballot64(o_t + len <= kOwnerBytes);
        is_tok = __builtin_amdgcn_inverse_ballot_w64(T);
        const unsigned last = 63u - (unsigned)__builtin_clzll(T);
        const unsigned N = (unsigned)__builtin_amdgcn_readlane(incl, (int)last);
        const unsigned adv = last + (unsigned)__builtin_amdgcn_readlane((int)tokbytes, (int)last);
        // validity of every element of the pass
        const bool bad = is_tok && ((kind != 0 && (off == 0 || off > op + o_t || (SLIDE && off > RING - 1024u))) ||
                                    len > out_len - op - o_t || o_t > out_len - op || ((len | off) & (GRAN - 1u)));
        if (__ballot(bad) != 0) { failed = true; break; }

        // ---- 4. owner map + per-byte production ----
        const unsigned NU = N / GRAN;                 // production units (bytes or 16-bit words)
        for (unsigned i = lane * 4u; i < NU; i += 256u)
            *reinterpret_cast<uint32_t *>(owner + i) = 0u;
        if (is_tok)
            owner[o_t / GRAN] = (uint8_t)(lane + 1u);
        // element attributes, fetched by the byte lanes with ds_bpermute
        // a0: o_t (11 bits) | len (9) << 11 | literal flag << 20 | offset saturated at 127 << 21 -- the offset
        // only matters for overlapping copies (off < len <= 64); a1: literal input coordinate or copy source
        // position
        const int a0 = (int)(o_t | (len << 11) | (kind == 0 ? (1u << 20) : 0u) | (min(off, 127u) << 21));
        const int a1 = (int)(kind == 0 ? x + hdr : op + o_t - off);
        unsigned carry = 0;
        const unsigned opu = op / GRAN;
        for (unsigned B = 0; B < NU; B += 64u) {
            const unsigned b = B + lane;
            const bool active = b < NU;
            int m = active ? (int)owner[b] : 0;
            m = wave_scan_max(m);
            m = max(m, (int)carry);
            carry = (unsigned)__builtin_amdgcn_readlane(m, 63);
            const unsigned sl = (unsigned)(m - 1) & 63u;
            const unsigned g0 = (unsigned)lane_gather(a0, sl);
            const unsigned g1 = (unsigned)lane_gather(a1, sl);
            const unsigned g2 = g0 >> 21;                                  // offset, saturated at 127
            const unsigned rel = b - (g0 & 0x7FFu) / GRAN;
            const unsigned elen = ((g0 >> 11) & 0x1FFu) / GRAN;
            const bool lit = ((g0 >> 20) & 1u) != 0;
            unsigned desc;                       // bit 31: resolved (LDS byte address), else output position
            // (both kinds are worked out side by side and selected: a branch on `lit` diverges in most steps)
            const unsigned desc_lit = 0x80000000u | (GRAN >= 2 ? 0x40000000u : 0u) | (RING + ((g1 + GRAN * rel) & (kV2InBytes - 1)));
            {
                unsigned r = rel;
                const unsigned offu = g2 / GRAN;
                if (!lit && offu < elen) {                       // overlapping copy: periodic pattern
                    // rel / offu, exact for rel < 256 and offu < 128: the reciprocal from v_rcp_f32 is exact enough
                    // (65536 / offu is an integer or at least 1/127 away from one)
                    const unsigned m = (unsigned)(65536.0f * __builtin_amdgcn_rcpf((float)offu)) + 1u;
                    const unsigned q = __umul24(rel, m) >> 16;
                    r = rel - __umul24(q, offu);
                }
                const unsigned q = g1 / GRAN + r;                // output position (in units) of the source
                desc = q < opu + B ? (0x80000000u | ((GRAN * q) & (RING - 1))) : q;
                // a stream's ring holds only the last RING bytes: older sources come back from memory.
                // The marker travels with the descriptor so that bytes copying from this byte inside
                // the same step inherit it (STREAM implies GRAN == 1, bit 30 is free there).
                if (STREAM && q + RING < op + kOwnerBytes + 64u)
                    desc = 0xC0000000u | q;
            }
            desc = lit ? desc_lit : desc;
            if (!active)
                desc = 0x80000000u;
            // sources inside this 64-byte step: pointer jumping
            if (ballot64((int)desc >= 0) != 0ull) {
                int round = 0;
                do {
                    const bool pending = (int)desc >= 0;
                    const unsigned from = (desc - (opu + B)) & 63u;
                    const unsigned g = (unsigned)lane_gather((int)desc, from);
                    desc = pending ? g : desc;
                } while (++round < 7 && ballot64((int)desc >= 0) != 0ull);
            }
            const unsigned a0 = desc & 0x3FFFFFFFu;
            const bool far = STREAM && (desc & 0x40000000u) != 0;
            unsigned value = far ? 0u : (unsigned)smem[a0];
            if (GRAN == 2) {
                // second byte: literals live in the 2 KiB staging ring (may wrap), copies are contiguous
                const unsigned a1 = (desc & 0x40000000u) ? RING + ((a0 - RING + 1u) & (kV2InBytes - 1)) : a0 + 1u;
                value |= (unsigned)smem[a1] << 8;
            }
            if (GRAN == 4) {
                // literals: 4 bytes at any byte position of the staging ring (two dwords + byte align);
                // copies: one aligned dword of the output ring
                const unsigned xs = a0 - RING;
                const unsigned w0 = inw[(xs >> 2) & (kV2InBytes / 4u - 1u)], w1 = inw[((xs >> 2) + 1u) & (kV2InBytes / 4u - 1u)];
                const unsigned lit4 = __builtin_amdgcn_alignbyte(w1, w0, xs & 3u);
                const unsigned cpy4 = *reinterpret_cast<const uint32_t *>(smem + ((desc & 0x40000000u) ? 0u : a0));
                value = (desc & 0x40000000u) ? lit4 : cpy4;
            }
            if (STREAM && __ballot(far) != 0) {
                // far sources were written to memory by THIS wave at least RING - 12 KiB of output ago:
                // drain the wave's own stores, then read back past the CU's L1 (sc1), which may still
                // hold the line as it was before those stores
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (far)
                    value = __hip_atomic_load(dst + a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (active) {
                if (GRAN == 4)
                    *reinterpret_cast<uint32_t *>(ring + ((op + 4u * b) & (RING - 1))) = value;
                else if (GRAN == 2)
                    *reinterpret_cast<uint16_t *>(ring + ((op + 2u * b) & (RING - 1))) = (uint16_t)value;
                else
                    ring[(op + b) & (RING - 1)] = (uint8_t)value;
            }
        }
        op += N;
        ip += adv;
        if (kFlushEarly && op - flushed >= kFlushAt) {
            const unsigned upto = op & ~(kFlushSegment - 1);
            flush_ring<RING>(ring, dst, flushed, upto, lane);
            flushed = upto;
        }
    }
    if (!failed && op != out_len)
        failed = true;
    if (failed) {
        if (lane == 0) {
            // (an 8 KiB BLOCK unit that fails -- the marks do not promise independent pieces -- hands the stream to the
            // second phase's units; a 64 KiB one -- a copy reaching before its block -- has the frame decoded again whole)
            if (fine_unit && phase == 1u) {
                HapGpuScanChunk *scan = (HapGpuScanChunk *)u.aux;
                __hip_atomic_store(&scan->fine_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            const unsigned code = (!STREAM || u.kind == HAPGPU_UNIT_SNAPPY_BLOCK)
                                      ? HAPGPU_STATUS_INDEX_MISMATCH
                                      : (job->mode == HAPGPU_JOB_SNAPPY ? kResInternal : kResBadFrame);
            atomicCAS(&job->status, 0u, code);
        }
        return;
    }
    if (op > flushed)
        flush_ring<RING>(ring, dst, flushed, op, lane);
}


// ------------------------------------------------------------------------------------------
// block scan for streams from other encoders
// ------------------------------------------------------------------------------------------
//
// libsnappy -- what the reference's HapEncode calls (hap.c:453) -- compresses independent 64 KiB blocks: no element
// crosses a multiple of 64 KiB of output and no copy reaches before the start of its block.  A chunk
// of such a stream (1.38 MB at 8K, 4 MiB at 16K) therefore is 22 / 64 independently decodable pieces -- if one knows
// where in the compressed bytes they begin.  Finding out means walking the element chain, which is serial; three
// kernels make it parallel, all built on the window parser of the decoder above (speculative parse of 64 byte
// positions, chain membership by pointer doubling) and none producing any output:
//
//   scan_walk_kernel   one wavefront per 4 KiB of COMPRESSED bytes (a "segment").  It starts 5 windows before its
//                      segment on a guess -- the element chain from an arbitrary byte joins the true chain within a
//                      few elements -- and records, for each of the segment's 64 windows, at which byte the chain
//                      entered it and how much output the chain had produced by then; plus where it left the segment.
//   scan_merge_kernel  one wavefront per chunk follows the TRUE chain from the stream's first element, but only until
//                      it stands on a byte the segment's record also entered: from there the record is the true
//                      chain, and the walk continues at the record's exit in the next segment (where, thanks to the
//                      warm-up, it usually matches at once).  This gives every segment its absolute output position.
//   scan_find_kernel   one wavefront per segment looks up which of its windows holds each multiple of 64 KiB of
//                      output, parses that one window again and notes the compressed position of the element that
//                      starts exactly there (none: an element straddles the boundary, the stream is not splittable).
//
// When every boundary of a chunk was found, its BLOCK units (written by the merge kernel into the slots the host
// reserved behind the stream unit) decode it, one wavefront per 64 KiB block, in the whole-stream kernel below with a
// small ring; otherwise they return at once and the stream unit runs as before.  What the scan does not check -- a
// copy reaching before its block -- the BLOCK unit's own offset check catches: the frame is then decoded again
// without the scan, as after a fragment table that lied.  An 8K frame's 24 chunks become 528 units.
//
// r04: marks every 8 KiB as well.  What plain hap.h HapEncode of THIS library writes has no table, but its chunks are
// concatenations of independent 8 KiB fragments: an element boundary at every 8 KiB of output, 4056 units per 8K
// frame instead of 528.  bpos[] is indexed by 8 KiB mark when the host reserved slots for them (a 64 KiB block begins
// at every eighth); the find kernel runs twice -- pass 0 takes the 64 KiB marks and the first two 8 KiB marks (the
// probe), pass 1 the other 8 KiB marks, only for streams whose probe marks both fell on element boundaries (libsnappy's
// hardly ever do: its scan costs what it did); scan_decide_kernel lists the fine units of the streams whose marks were
// ALL found, and the decode launch runs that list as a first phase.  Marks on element boundaries do not make pieces
// independent: a fine unit that fails hands its stream to the 64 KiB blocks / the stream unit of the second phase.
constexpr unsigned kScanSegment = HAPGPU_SCAN_SEGMENT;       // the longest segment (LDS is sized for it); a call of few compressed bytes uses half
__device__ __forceinline__ unsigned scan_segment_bytes(const HapGpuScanChunk &sc)
{
    return (sc.seg_bytes == kScanSegment / 2u) ? kScanSegment / 2u : kScanSegment;
}
// (5 windows -- longer than the longest element that is not a "long literal", 258 bytes -- let libsnappy's streams
// join; the streams of this library's block compressor, with their long literal runs in noisy areas, need more: with 5
// the merge kernel walked enough windows itself to take 0.22 ms for one 8K frame, with 12 it takes 0.05)
constexpr unsigned kScanWarmWindows = 12u;
// (r06: calls of few segments warm up over twice as many windows.  The segments whose guess has not joined the true chain by
// the time it enters them are walked by scan_merge, ONE wavefront per stream: in a single plain 8K frame of this library 5 % of
// the segments -- the ones that begin in noise, where elements are 130-byte literals and a guess inside one hops through
// texture bytes -- cost the merge 833 windows, 139 us for the slowest stream, more than the walk itself (59 us).  A batch has
// streams enough to hide that, and pays for every warm-up window in the walk.)
constexpr unsigned kScanWarmWindowsShortCall = 24u;
constexpr unsigned kScanLdsFor(unsigned warm) { return kScanSegment + 64u * warm + 128u; }
constexpr unsigned kBlockOut = 65536u;
constexpr unsigned kFine = HAPGPU_SCAN_FINE;                  // 8 KiB: the fragments of this library's own streams
constexpr unsigned kRecNone = 0xFFu;
constexpr unsigned kMergeSegments = 2048u;                    // segments of one stream the merge kernel tabulates (8 MiB compressed)

// One 16-byte / 4-byte piece of the stream at aligned coordinate c, without touching bytes at or after in_end.
__device__ __forceinline__ uint4 scan_load16(const uint8_t *src_al, unsigned c, unsigned in_end)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c + 16u <= in_end) {
        v = *reinterpret_cast<const uint4 *>(src_al + c);
    } else if (c < in_end) {
        unsigned w[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (unsigned k = 0; k < 16u && c + k < in_end; k++)
            w[k >> 2] |= (unsigned)src_al[c + k] << (8 * (k & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

__device__ __forceinline__ unsigned scan_load4(const uint8_t *src_al, unsigned c, unsigned in_end)
{
    unsigned v = 0;
    if (c + 4u <= in_end) {
        v = *reinterpret_cast<const uint32_t *>(src_al + c);
    } else if (c < in_end) {
#pragma unroll 1
        for (unsigned k = 0; k < 4u && c + k < in_end; k++)
            v |= (unsigned)src_al[c + k] << (8 * k);
    }
    return v;
}

// The element that would start at coordinate x, from the five bytes there (lo: bytes x..x+3, hi: byte x+4).
struct ScanElement {
    unsigned len;        // output bytes
    unsigned tokbytes;   // input bytes
    bool stopper;        // not an element the window parser takes: a literal with 2..4 length bytes, or cut off by the input's end
};

__device__ __forceinline__ ScanElement scan_parse(unsigned lo, unsigned x, unsigned in_end)
{
    const unsigned tag = lo & 0xFFu, kind = tag & 3u;
    unsigned len, hdr;
    bool special = false;
    if (kind == 0) {
        len = (tag >> 2) + 1u;
        hdr = 1;
        if (len == 61u) {
            len = ((lo >> 8) & 0xFFu) + 1u;
            hdr = 2;
        } else if (len > 61u) {
            special = true;
            hdr = 1u + (len - 60u);
        }
    } else if (kind == 1) {
        len = 4u + ((tag >> 2) & 7u);
        hdr = 2;
    } else if (kind == 2) {
        len = (tag >> 2) + 1u;
        hdr = 3;
    } else {
        len = (tag >> 2) + 1u;
        hdr = 5;
    }
    ScanElement el;
    el.len = len;
    el.tokbytes = hdr + (kind == 0 ? len : 0u);
    el.stopper = special || x >= in_end || el.tokbytes > in_end - x;
    return el;
}

__device__ __forceinline__ unsigned long long scan_chain(const ScanElement &el, unsigned lane, unsigned from)
{
    return window_chain(el.stopper, el.stopper ? 64u : min(lane + el.tokbytes, 64u), lane, from);
}

// The long literal (2..4 length bytes) at coordinate p, from its first five bytes: header and payload sizes.
// false: not a long literal, or it does not fit the input.
__device__ __forceinline__ bool scan_long_literal(unsigned lo0, unsigned hi0, unsigned p, unsigned in_end, unsigned *hdr, unsigned *llen)
{
    const unsigned tag0 = lo0 & 0xFFu;
    if ((tag0 & 3u) != 0 || (tag0 >> 2) < 61u || p >= in_end)
        return false;
    const unsigned extra = (tag0 >> 2) - 59u;
    const unsigned long long field = (((unsigned long long)hi0 << 32) | lo0) >> 8;
    const unsigned v = (unsigned)(extra == 4 ? field : (field & ((1ull << (8 * extra)) - 1ull)));
    const unsigned h = 1u + extra;
    if (h > in_end - p || v == 0xFFFFFFFFu || v + 1u > in_end - p - h)
        return false;
    *hdr = h;
    *llen = v + 1u;
    return true;
}

// chunk of global segment g: the last chunk whose seg_first is <= g
__device__ __forceinline__ unsigned scan_chunk_of(const HapGpuScanChunk *chunks, unsigned chunk_count, unsigned g)
{
    unsigned lo = 0, hi = chunk_count;
    while (hi - lo > 1u) {
        const unsigned mid = (lo + hi) >> 1;
        if (chunks[mid].seg_first <= g)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__device__ __forceinline__ bool scan_unit_wanted(const HapGpuDecodeUnit &u, const HapGpuDecodeJob *jobs)
{
    return u.kind == HAPGPU_UNIT_SNAPPY_STREAM && u.aux != 0u && jobs[u.job].status == 0u && u.dst_len > kBlockOut;
}

__global__ __launch_bounds__(64) void scan_walk_kernel(const HapGpuDecodeUnit *units, const HapGpuDecodeJob *jobs,
                                                       const HapGpuScanChunk *chunks, unsigned chunk_count,
                                                       HapGpuScanSegment *__restrict__ segs, unsigned long long *__restrict__ recs,
                                                       uint4 *__restrict__ joins, unsigned seg_total, unsigned warm_windows)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];        // kScanLdsFor(warm_windows) bytes
    const uint32_t *inw = reinterpret_cast<const uint32_t *>(smem);
    const unsigned lane = threadIdx.x, g = blockIdx.x;
    if (g >= seg_total)
        return;
    const HapGpuScanChunk sc = chunks[scan_chunk_of(chunks, chunk_count, g)];
    const unsigned s = g - sc.seg_first;
    const HapGpuDecodeUnit u = units[sc.unit];
    const unsigned shift = (unsigned)(u.src & 15u);
    const uint8_t *src_al = (const uint8_t *)u.src - shift;
    const unsigned in_end = shift + u.src_len;
    const unsigned seg_bytes = scan_segment_bytes(sc);
    const unsigned seg_begin = s * seg_bytes, seg_end = seg_begin + seg_bytes;
    unsigned long long rec = kRecNone;
    unsigned flags = 1u, cum = 0, cum_e = 0, p = 0;           // (cum_e: elements, as cum counts their output bytes)
    if (scan_unit_wanted(u, jobs) && s < sc.seg_count && seg_begin < in_end) {
        const unsigned warm = s == 0 ? 0u : warm_windows;
        const unsigned stage_begin = seg_begin - 64u * warm;
        const unsigned stage_end = min(seg_end + 64u, (in_end + 15u) & ~15u);
        for (unsigned c = stage_begin + lane * 16u; c < stage_end; c += 1024u)
            *reinterpret_cast<uint4 *>(smem + (c - stage_begin)) = scan_load16(src_al, c, in_end);
        __syncthreads();
        p = stage_begin;
        if (s == 0) {      // the stream's first element follows its length prefix (validated by the plan kernel)
            unsigned b;
            p = shift;
            do {
                b = smem[p];
                p++;
            } while ((b & 0x80u) && p < in_end);
            p = uniform(p);
        }
        flags = 0;
        unsigned seen = 0xFFFFFFFFu;            // window whose entry is on record
        while (p < seg_end) {
            if (p >= in_end) {
                flags = 2u;                      // (advances never pass in_end)
                break;
            }
            const unsigned wi = (p - stage_begin) >> 6, ws = stage_begin + wi * 64u, e = p - ws;
            if (wi != seen && wi >= warm && lane == wi - warm)
                rec = ((unsigned long long)cum_e << 40) | ((unsigned long long)cum << 8) | e;
            seen = wi;
            const unsigned x = ws + lane, xi = (x - stage_begin) >> 2, sh = x & 3u;
            const unsigned w0 = inw[xi], w1 = inw[xi + 1u];
            const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
            const unsigned hi = (w1 >> (8u * sh)) & 0xFFu;
            const ScanElement el = scan_parse(lo, x, in_end);
            const unsigned long long T = scan_chain(el, lane, e);
            if (T == 0ull) {
                unsigned h = 0, llen = 0;
                const bool is_long = scan_long_literal((unsigned)__builtin_amdgcn_readlane((int)lo, (int)e),
                                                       (unsigned)__builtin_amdgcn_readlane((int)hi, (int)e), p, in_end, &h, &llen);
                if (p < seg_begin) {             // still guessing: try again from the next window
                    p = ws + 64u;
                    continue;
                }
                if (!is_long) {
                    flags = 1u;                  // not an element: if this is the true chain, the stream is broken
                    break;
                }
                p += h + llen;
                cum += llen;
                cum_e += 1u;
                continue;
            }
            const bool is_tok = __builtin_amdgcn_inverse_ballot_w64(T);
            const int incl = wave_scan_add(is_tok ? (int)el.len : 0);
            const unsigned last = 63u - (unsigned)__builtin_clzll(T);
            cum += (unsigned)__builtin_amdgcn_readlane(incl, 63);
            cum_e += (unsigned)__builtin_popcountll(T);
            p = ws + last + (unsigned)__builtin_amdgcn_readlane((int)el.tokbytes, (int)last);
        }
        if (flags == 0u && p >= in_end)
            flags = 2u;
    }
    recs[(size_t)g * 64u + lane] = rec;
    if (lane == 0) {
        HapGpuScanSegment sg;
        sg.exit_coord = p;
        sg.cum_total = cum;
        sg.flags = flags;
        sg.elements = cum_e;
        segs[g] = sg;
        joins[g] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);  // (window, output position and element number of the record's zero, late): not joined yet
    }
}

// A window read straight from memory (the merge and find kernels visit few windows).
__device__ __forceinline__ void scan_window_bytes(const uint8_t *src_al, unsigned x, unsigned in_end, unsigned *lo, unsigned *hi)
{
    const unsigned c = x & ~3u, sh = x & 3u;
    const unsigned w0 = scan_load4(src_al, c, in_end), w1 = scan_load4(src_al, c + 4u, in_end);
    *lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    *hi = (w1 >> (8u * sh)) & 0xFFu;
}

#ifdef BRK_TIMING
__device__ unsigned g_merge_dbg[8];       // [0] streams, [1] segments, [2] good segments, [3] windows parsed here, [4] record joins, [5] clock ticks
#define MERGE_DBG(k, v) do { if (lane == 0) atomicAdd(&g_merge_dbg[k], (unsigned)(v)); } while (0)
#else
#define MERGE_DBG(k, v)
#endif
__global__ __launch_bounds__(64) void scan_merge_kernel(HapGpuDecodeUnit *units, const HapGpuDecodeJob *jobs,
                                                        HapGpuScanChunk *chunks, unsigned chunk_count,
                                                        const HapGpuScanSegment *__restrict__ segs,
                                                        const unsigned long long *__restrict__ recs, uint4 *__restrict__ joins,
                                                        uint32_t *fine_cursor, unsigned fine_first, unsigned fine_pool)
{
    const unsigned lane = threadIdx.x, c = blockIdx.x;
    if (c >= chunk_count)
        return;
    const HapGpuScanChunk sc = chunks[c];
    const HapGpuDecodeUnit u = units[sc.unit];
    const unsigned out_len = u.dst_len;
    const unsigned nblk = (out_len + kBlockOut - 1u) / kBlockOut;
    const unsigned nfine = (out_len + kFine - 1u) / kFine;
    if (!scan_unit_wanted(u, jobs) || nblk > sc.slots)
        return;                                   // (ok stays 0: the host sent zeros)
    // marks every 8 KiB of output when the host reserved a unit slot for each (else every 64 KiB, as for any stream)
    bool fine_on = sc.fine_slots != 0u && nfine <= sc.fine_slots && fine_cursor != nullptr;
    // The 8 KiB blocks' unit slots come out of a pool behind the call's ordinary units, handed out here: the host knows
    // what all the streams of a frame can produce together (its texture), not how the chunks share it (r05: a slot range
    // per stream sized by what ITS bytes could expand to made the pool of 60 8K frames of 24 chunks 2.07 M slots for
    // 243 000 blocks -- and two launches of 2 M wavefronts that found nothing to do, 0.7 ms).  A stream that finds the
    // pool empty -- lengths that lie -- goes on with its 64 KiB blocks.
    unsigned fine_at = 0;
    if (fine_on) {
        if (lane == 0)
            fine_at = atomicAdd(fine_cursor, nfine);
        fine_at = uniform(fine_at);
        fine_on = fine_at <= fine_pool && nfine <= fine_pool - fine_at;
        fine_at += fine_first;
    }
    const unsigned mark = fine_on ? kFine : kBlockOut;
    const unsigned shift = (unsigned)(u.src & 15u);
    const uint8_t *src_al = (const uint8_t *)u.src - shift;
    const unsigned in_end = shift + u.src_len;
    uint32_t *bpos = (uint32_t *)sc.bpos;
    unsigned p = shift;
    {
        unsigned b;
        do {
            b = src_al[p];
            p++;
        } while ((b & 0x80u) && p < in_end);
        p = uniform(p);
    }
    unsigned op = 0, oe = 0, found = 0, found_fine = 0, found_probe = 0, cur = 0xFFFFFFFFu;   // (oe: elements so far, as op counts output bytes)
    unsigned long long rec = kRecNone, rec_ahead = kRecNone;
    HapGpuScanSegment sg = {}, sg_ahead = {};
    bool ok = true;
    // The usual case, worked out for all segments at once (one lane per segment): the chain recorded for segment i - 1
    // leaves it inside segment i at a byte where the chain recorded for segment i entered a window (thanks to the
    // warm-up it has joined the true chain by then) -- segment i is "good".  Along a run of good segments the recorded
    // chains are the true chain and the output positions are prefix sums; the walk below takes such runs in one step and
    // parses windows itself only in the segments that are not good (about one in a hundred).
    __shared__ uint32_t l_exit[kMergeSegments];             // where segment i's recorded chain leaves it
    __shared__ uint32_t l_before[kMergeSegments + 1u];      // output of the good segments before i (entry to exit each)
    __shared__ uint32_t l_before_e[kMergeSegments + 1u];    // ... and their elements
    __shared__ unsigned long long l_good[kMergeSegments / 64u];
    const unsigned seg_bytes = scan_segment_bytes(sc);
    const unsigned nseg = (in_end + seg_bytes - 1u) / seg_bytes;
    const unsigned first_element = p;
    const bool tabled = nseg <= sc.seg_count && nseg <= kMergeSegments;
#ifdef BRK_TIMING
    const unsigned long long t_begin = wall_clock64();
#endif
    MERGE_DBG(0, 1);
    MERGE_DBG(1, nseg);
    if (tabled) {
        unsigned run = 0, run_e = 0;
        for (unsigned base = 0; base < nseg; base += 64u) {
            const unsigned i = base + lane;
            const bool mine = i < nseg;
            unsigned entry = first_element, flags_prev = 0;
            if (mine && i > 0u) {
                const HapGpuScanSegment prev = segs[sc.seg_first + i - 1u];
                entry = prev.exit_coord;
                flags_prev = prev.flags;
            }
            HapGpuScanSegment here = {};
            unsigned long long r = kRecNone;
            if (mine)
                here = segs[sc.seg_first + i];
            const bool inside = mine && flags_prev == 0u && entry / seg_bytes == i && entry < in_end;
            if (inside)
                r = recs[(size_t)(sc.seg_first + i) * 64u + ((entry % seg_bytes) >> 6)];
            const bool good = inside && ((unsigned)r & 0xFFu) == (entry & 63u) && (here.flags & 1u) == 0u;
            const unsigned delta = good ? here.cum_total - (unsigned)(r >> 8) : 0u;
            const unsigned delta_e = good ? here.elements - (unsigned)(r >> 40) : 0u;
            const unsigned incl = (unsigned)wave_scan_add((int)delta);
            const unsigned incl_e = (unsigned)wave_scan_add((int)delta_e);
            if (mine) {
                l_exit[i] = here.exit_coord;
                l_before[i] = run + incl - delta;
                l_before_e[i] = run_e + incl_e - delta_e;
            }
            const unsigned long long goods = ballot64(good);
            MERGE_DBG(2, __builtin_popcountll(goods));
            if (lane == 0)
                l_good[base / 64u] = goods;
            run += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            run_e += (unsigned)__builtin_amdgcn_readlane((int)incl_e, 63);
        }
        if (lane == 0) {
            l_before[nseg] = run;
            l_before_e[nseg] = run_e;
        }
        __syncthreads();
    }
    while (p < in_end) {
        const unsigned s = p / seg_bytes, wi = (p % seg_bytes) >> 6, e = p & 63u;
        if (tabled && p == (s == 0u ? first_element : l_exit[s - 1u]) && ((l_good[s / 64u] >> (s & 63u)) & 1ull)) {
            // a run of good segments [s, t): all their joins at once, then on to where the last one's chain leaves
            unsigned t = s + 1u;
            while (t < nseg && ((l_good[t / 64u] >> (t & 63u)) & 1ull))
                t++;
            const unsigned before_run = l_before[s], before_run_e = l_before_e[s];
            for (unsigned i = s + lane; i < t; i += 64u) {
                const unsigned entry = i == 0u ? first_element : l_exit[i - 1u];
                const unsigned w = (entry % seg_bytes) >> 6;
                const unsigned long long r = recs[(size_t)(sc.seg_first + i) * 64u + w];
                // (w = 0 in the last word: the chain enters no window of this segment in front of the one on record)
                joins[sc.seg_first + i] = make_uint4(w, op + (l_before[i] - before_run) - (unsigned)(r >> 8),
                                                     oe + (l_before_e[i] - before_run_e) - (unsigned)(r >> 40), 0u);
            }
            op += l_before[t] - before_run;
            oe += l_before_e[t] - before_run_e;
            p = l_exit[t - 1u];
            cur = 0xFFFFFFFFu;
            continue;
        }
        if (s != cur) {
            // the next segment's record was asked for when this one was entered (the chain nearly always goes there)
            if (s == cur + 1u && cur != 0xFFFFFFFFu) {
                rec = rec_ahead;
                sg = sg_ahead;
            } else {
                rec = recs[(size_t)(sc.seg_first + s) * 64u + lane];
                sg = segs[sc.seg_first + s];
            }
            cur = s;
            if (s + 1u < sc.seg_count) {
                rec_ahead = recs[(size_t)(sc.seg_first + s + 1u) * 64u + lane];
                sg_ahead = segs[sc.seg_first + s + 1u];
            }
        }
        const unsigned r_lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rec, (int)wi);
        const unsigned r_hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rec >> 32), (int)wi);
        if ((r_lo & 0xFFu) == e) {
            // the record's chain entered this window where the true chain stands: the rest of the segment is on record
            if (sg.flags & 1u) {
                ok = false;
                break;
            }
            MERGE_DBG(4, 1);
            const unsigned at_entry = (r_lo >> 8) | (r_hi << 24);
            const unsigned base_op = op - at_entry, base_e = oe - (r_hi >> 8);
            // (last word: 1 = the true chain may have entered this segment in front of window wi -- those windows' records
            // are not its own)
            if (lane == 0)
                joins[sc.seg_first + s] = make_uint4(wi, base_op, base_e, 1u);
            p = sg.exit_coord;
            op = base_op + sg.cum_total;
            oe = base_e + sg.elements;
            continue;
        }
        // not yet: one window of the true chain, parsed here
        MERGE_DBG(3, 1);
        const unsigned ws = p - e, x = ws + lane;
        unsigned lo, hi;
        scan_window_bytes(src_al, x, in_end, &lo, &hi);
        const ScanElement el = scan_parse(lo, x, in_end);
        const unsigned long long T = scan_chain(el, lane, e);
        if (T == 0ull) {
            unsigned h = 0, llen = 0;
            if (!scan_long_literal((unsigned)__builtin_amdgcn_readlane((int)lo, (int)e),
                                   (unsigned)__builtin_amdgcn_readlane((int)hi, (int)e), p, in_end, &h, &llen)) {
                ok = false;
                break;
            }
            if ((op & (mark - 1u)) == 0u && op < out_len) {
                if (lane == 0)
                    bpos[op / mark] = p;
                found_fine += 1u;
                found += (op & (kBlockOut - 1u)) == 0u ? 1u : 0u;
                found_probe += (op == kFine || op == 2u * kFine) ? 1u : 0u;
            }
            p += h + llen;
            op += llen;
            oe += 1u;
            continue;
        }
        const bool is_tok = __builtin_amdgcn_inverse_ballot_w64(T);
        const int incl = wave_scan_add(is_tok ? (int)el.len : 0);
        oe += (unsigned)__builtin_popcountll(T);
        const unsigned at = op + (unsigned)incl - (is_tok ? el.len : 0u);
        const bool starts_mark = is_tok && (at & (mark - 1u)) == 0u && at < out_len;
        if (starts_mark)
            bpos[at / mark] = x;
        found_fine += (unsigned)__builtin_popcountll(ballot64(starts_mark));
        found += (unsigned)__builtin_popcountll(ballot64(starts_mark && (at & (kBlockOut - 1u)) == 0u));
        found_probe += (unsigned)__builtin_popcountll(ballot64(starts_mark && (at == kFine || at == 2u * kFine)));
        const unsigned last = 63u - (unsigned)__builtin_clzll(T);
        op += (unsigned)__builtin_amdgcn_readlane(incl, 63);
        p = ws + last + (unsigned)__builtin_amdgcn_readlane((int)el.tokbytes, (int)last);
        if (op > out_len) {
            ok = false;
            break;
        }
    }
#ifdef BRK_TIMING
    MERGE_DBG(5, wall_clock64() - t_begin);
#endif
    if (!ok || p != in_end || op != out_len)
        return;
    HapGpuScanChunk *state = chunks + c;
    // the 64 KiB blocks' unit slots follow the stream unit; the 8 KiB blocks' lie behind all ordinary units of the call
    for (unsigned b = lane; fine_on && b < nfine; b += 64u) {
        HapGpuDecodeUnit w;
        w.src = (uint64_t)src_al;
        w.dst = u.dst + (uint64_t)b * kFine;
        w.src_len = 0;
        w.dst_len = min(kFine, out_len - b * kFine);
        w.kind = HAPGPU_UNIT_SNAPPY_BLOCK;
        w.job = u.job;
        w.aux = (uint64_t)state;
        w.reserved = (uint64_t)b | HAPGPU_BLOCK_FINE;
        units[fine_at + b] = w;
    }
    for (unsigned b = lane; b < nblk; b += 64u) {
        HapGpuDecodeUnit w;
        w.src = (uint64_t)src_al;
        w.dst = u.dst + (uint64_t)b * kBlockOut;
        w.src_len = 0;
        w.dst_len = min(kBlockOut, out_len - b * kBlockOut);
        w.kind = HAPGPU_UNIT_SNAPPY_BLOCK;
        w.job = u.job;
        w.aux = (uint64_t)state;
        w.reserved = b;
        units[sc.unit + 1u + b] = w;
    }
    if (lane == 0) {
        bpos[fine_on ? nfine : nblk] = in_end;
        units[sc.unit].reserved = (uint64_t)state;
        state->expected = nblk;
        state->expected_fine = fine_on ? nfine : 0u;
        state->fine_unit_first = fine_at;
        atomicAdd(&state->found, found);
        atomicAdd(&state->found_fine, found_fine);
        atomicAdd(&state->probe_found, found_probe);
        state->ok = 1u;
    }
}

// pass 0: the 64 KiB marks, and of the 8 KiB marks the first two (the probe); pass 1: the other 8 KiB marks, for the
// streams whose probe marks both fell on element boundaries
__global__ __launch_bounds__(64) void scan_find_kernel(const HapGpuDecodeUnit *units, HapGpuScanChunk *chunks, unsigned chunk_count,
                                                       const HapGpuScanSegment *__restrict__ segs,
                                                       const unsigned long long *__restrict__ recs, const uint4 *__restrict__ joins,
                                                       unsigned seg_total, unsigned pass)
{
    const unsigned lane = threadIdx.x, g = blockIdx.x;
    if (g >= seg_total)
        return;
    const unsigned c = scan_chunk_of(chunks, chunk_count, g);
    const HapGpuScanChunk sc = chunks[c];
    const HapGpuScanSegment sg = segs[g];
    const uint4 join = joins[g];
    const unsigned merge_window = join.x, base_op = join.y;
    if (!sc.ok || merge_window >= 64u)
        return;
    if (pass == 1u && (sc.expected_fine == 0u || sc.probe_found != min(2u, sc.expected_fine - 1u)))
        return;
    const HapGpuDecodeUnit u = units[sc.unit];
    const unsigned shift = (unsigned)(u.src & 15u);
    const uint8_t *src_al = (const uint8_t *)u.src - shift;
    const unsigned in_end = shift + u.src_len, out_len = u.dst_len;
    uint32_t *bpos = (uint32_t *)sc.bpos;
    const unsigned seg_begin = (g - sc.seg_first) * scan_segment_bytes(sc);
    const unsigned long long rec = recs[(size_t)g * 64u + lane];
    const unsigned entry = (unsigned)rec & 0xFFu;
    const unsigned abs_op = base_op + (unsigned)(rec >> 8);              // output position at this window's entry (bits 8..39)
    const bool usable = lane >= merge_window && entry != kRecNone;
    const unsigned first_op = (unsigned)__builtin_amdgcn_readlane((int)abs_op, (int)merge_window);
    const unsigned exit_op = base_op + sg.cum_total;
    unsigned found = 0, found_fine = 0;
    // (the merge kernel, which ran before, decided the granularity of the marks)
    const unsigned mark = sc.expected_fine != 0u ? kFine : kBlockOut;
    // marks V with first_op <= V < exit_op belong to elements that start in this segment's recorded windows
    unsigned found_probe = 0;
    for (unsigned long long V = ((unsigned long long)first_op + mark - 1u) / mark * mark;
         V < exit_op && V < out_len; V += mark) {
        // (with 8 KiB marks: pass 0 takes the multiples of 64 KiB and the two probe marks, pass 1 the others)
        const bool early = (V & (kBlockOut - 1u)) == 0ull || V == kFine || V == 2u * kFine;
        if (mark == kFine && early != (pass == 0u))
            continue;
        const bool is_probe = mark == kFine && (V == kFine || V == 2u * kFine);
        const unsigned long long m = ballot64(usable && abs_op <= (unsigned)V);
        if (m == 0ull)
            continue;
        const unsigned k = 63u - (unsigned)__builtin_clzll(m);          // the last window entered at or before V
        const unsigned ws = seg_begin + 64u * k;
        unsigned p = ws + (unsigned)__builtin_amdgcn_readlane((int)entry, (int)k);
        unsigned op = (unsigned)__builtin_amdgcn_readlane((int)abs_op, (int)k);
        const unsigned x = ws + lane;
        unsigned lo, hi;
        scan_window_bytes(src_al, x, in_end, &lo, &hi);
        const ScanElement el = scan_parse(lo, x, in_end);
        while (p < ws + 64u && p < in_end && op <= (unsigned)V) {
            const unsigned e = p - ws;
            const unsigned long long T = scan_chain(el, lane, e);
            if (T == 0ull) {
                unsigned h = 0, llen = 0;
                if (!scan_long_literal((unsigned)__builtin_amdgcn_readlane((int)lo, (int)e),
                                       (unsigned)__builtin_amdgcn_readlane((int)hi, (int)e), p, in_end, &h, &llen))
                    break;
                if (op == (unsigned)V) {
                    if (lane == 0)
                        bpos[V / mark] = p;
                    found_fine += 1u;
                    found += (V & (kBlockOut - 1u)) == 0ull ? 1u : 0u;
                    found_probe += is_probe ? 1u : 0u;
                }
                p += h + llen;
                op += llen;
                continue;
            }
            const bool is_tok = __builtin_amdgcn_inverse_ballot_w64(T);
            const int incl = wave_scan_add(is_tok ? (int)el.len : 0);
            const unsigned at = op + (unsigned)incl - (is_tok ? el.len : 0u);
            const bool hit = is_tok && at == (unsigned)V;
            if (hit)
                bpos[V / mark] = x;
            const unsigned hits = (unsigned)__builtin_popcountll(ballot64(hit));
            found_fine += hits;
            found += (V & (kBlockOut - 1u)) == 0ull ? hits : 0u;
            found_probe += is_probe ? hits : 0u;
            const unsigned last = 63u - (unsigned)__builtin_clzll(T);
            op += (unsigned)__builtin_amdgcn_readlane(incl, 63);
            p = ws + last + (unsigned)__builtin_amdgcn_readlane((int)el.tokbytes, (int)last);
        }
    }
    if (lane == 0 && found)
        atomicAdd(&chunks[c].found, found);
    if (lane == 0 && found_fine)
        atomicAdd(&chunks[c].found_fine, found_fine);
    if (lane == 0 && found_probe)
        atomicAdd(&chunks[c].probe_found, found_probe);
}

// After the find passes: the streams whose 8 KiB marks were all found put their fine units on the list the decode
// launch's first phase works through (one wavefront per stream).
__global__ __launch_bounds__(64) void scan_decide_kernel(const HapGpuScanChunk *chunks, unsigned chunk_count, uint32_t *work)
{
    const unsigned lane = threadIdx.x, c = blockIdx.x;
    if (c >= chunk_count)
        return;
    const HapGpuScanChunk sc = chunks[c];
    if (!sc.ok || sc.expected_fine == 0u || sc.found_fine != sc.expected_fine || sc.expected_fine > sc.fine_slots)
        return;
    unsigned base = 0;
    if (lane == 0)
        base = atomicAdd(&work[0], sc.expected_fine);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    for (unsigned b = lane; b < sc.expected_fine; b += 64u)
        work[1u + base + b] = sc.fine_unit_first + b;
}


// ------------------------------------------------------------------------------------------
// 64 KiB blocks of other encoders' streams, a WORKGROUP each: pointers first, bytes last
// ------------------------------------------------------------------------------------------
//
// The kernel above decodes a libsnappy block (hap.c:606-642's snappy_uncompress, for frames the reference's own HapEncode
// wrote, hap.c:448-476) with ONE wavefront that takes its ~330 windows of 64 compressed bytes one after the other: 0.8 ms
// for a block however idle the GPU is, and a whole 8K frame is only 506 blocks.  What is serial about a Snappy stream is
// (a) where its elements begin and (b) that copies read what earlier elements wrote.  (a) is on record: the block scan
// left, for every window of compressed bytes, the byte at which the element chain enters it and the output position
// there (scan_walk's records, joined to the true chain by scan_merge).  (b) is a forest: every output byte either IS a
// literal byte or EQUALS one earlier output byte of its block.  So sixteen wavefronts take a block together:
//
//   A. every window with a record is parsed speculatively (all 64 byte positions, chain membership by pointer doubling
//      from the recorded entry -- the parser of the kernels above) and notes where its chain leaves it;
//   B. the records are VERIFIED, not trusted: each window's exit must be the recorded entry (byte and output position)
//      of the window it lands in, the block's first window is entered at the block's mark, exactly one chain ends at the
//      next mark with the block's length, and no window is entered twice.  Windows the true chain enters without a usable
//      record (the start of the one segment in a hundred whose guessed chain had not joined the true one yet) are
//      walked by one wavefront from the exit of the window before them;
//   C. the windows are parsed again, now with verified entries, and every output byte gets a 16-bit POINTER in LDS:
//      a literal byte points to itself (and goes to memory at once), a copy byte to the byte it copies -- inside an
//      overlapping copy to the pattern's first period, so that an element never chains through itself;
//   D. pointer jumping, ptr[i] = ptr[ptr[i]], until nothing moves: as many rounds as the logarithm of the longest
//      chain of copies of copies (3 to 5 for Hap textures), every round over all 64 Ki pointers by all 1024 lanes;
//   E. every copy byte fetches the literal byte its pointer names from memory and is stored.
//
// Anything unexpected -- a record that does not verify, an element the window parser does not take, a copy that reaches
// before its block -- and the kernel returns without a word: the unit is still there for the wavefront-per-block kernel
// of the launch that follows, which decides what the stream's fault is called.  A block that went through becomes a SKIP
// unit.  One 8K frame of the reference encoder: two blocks per CU instead of two wavefronts per CU.
constexpr unsigned kBrkWaves = 16u, kBrkThreads = 64u * kBrkWaves;
constexpr unsigned kBrkMaxWindows = 1024u;          // compressed bytes of a block this kernel takes: 64 KiB (libsnappy's blocks: ~21 KiB of a Hap Q texture)
constexpr unsigned kBrkOwner = 512u;                // output bytes of one production pass of a wavefront
constexpr unsigned kBrkStage = 512u;                // compressed bytes a wavefront stages per window: the window and the longest short literal behind it
constexpr unsigned kBrkMaxRounds = 18u;

struct BrkLds {
    uint16_t ptr[kBlockOut];
    uint32_t w_op[kBrkMaxWindows];                  // output position (inside the block) at the window's entry
    uint32_t w_exit_p[kBrkMaxWindows];              // where the chain leaves the window (stream coordinate)
    uint32_t w_exit_op[kBrkMaxWindows];             // ... and the output position there
    uint8_t w_entry[kBrkMaxWindows];                // byte of the window at which the chain enters it; 0xFF: not on record
    uint8_t w_in[kBrkMaxWindows];                   // windows whose chain leaves into this one
    uint8_t owner[kBrkWaves][kBrkOwner];
    uint8_t stage[kBrkWaves][kBrkStage + 16u];
    uint32_t fail;
    uint32_t ends;                                  // chains that end at the block's end
    uint32_t work;                                  // the block the workgroup has just taken
    uint32_t stream_first[256u + 1u];               // blocks of the call's scanned streams before stream c (kBrkMaxStreams)
};

// eight bytes of the stream at the 8-byte aligned coordinate c; nothing at or beyond `end` is touched
__device__ __forceinline__ uint2 brk_load8(const uint8_t *src_al, unsigned c, unsigned end)
{
    uint2 v = make_uint2(0u, 0u);
    if (c + 8u <= end) {
        v = *reinterpret_cast<const uint2 *>(src_al + c);
    } else if (c < end) {
        unsigned w[2] = {0u, 0u};
#pragma unroll 1
        for (unsigned k = 0; k < 8u && c + k < end; k++)
            w[k >> 2] |= (unsigned)src_al[c + k] << (8u * (k & 3u));
        v = make_uint2(w[0], w[1]);
    }
    return v;
}

// One window of one wavefront: the elements that begin in [ws + e, ws + 64) -- and the long literal the chain may stop at --
// get their pointers (and the literal bytes go to memory); where the chain leaves the window comes back.  `bytes`: the
// lane's eight of the 512 bytes from ws on.  false: something this kernel does not read.
__device__ __forceinline__ bool brk_do_window(BrkLds &L, unsigned wave, unsigned lane, const uint8_t *src_al, uint8_t *dst, unsigned ws,
                                              unsigned e, unsigned op, unsigned to, unsigned out_len, const uint2 bytes,
                                              unsigned *exit_p, unsigned *exit_op)
{
    uint8_t *stage = L.stage[wave], *owner = L.owner[wave];
    *reinterpret_cast<uint2 *>(stage + lane * 8u) = bytes;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // (stored as 8 bytes, read as dwords: no reordering by the compiler)
    const unsigned x = ws + lane;
    const uint32_t *st32 = reinterpret_cast<const uint32_t *>(stage);
    const unsigned w0 = st32[lane >> 2], w1 = st32[(lane >> 2) + 1u];      // (a wave's LDS accesses complete in order)
    const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, lane);          // bytes x .. x + 3 (the shift is lane & 3)
    const unsigned hi = (w1 >> (8u * (lane & 3u))) & 0xFFu;                // byte x + 4
    const unsigned tag = lo & 0xFFu, kind = tag & 3u;
    unsigned len, off = 0, hdr;
    bool special = false;
    if (kind == 0u) {
        len = (tag >> 2) + 1u;
        hdr = 1u;
        if (len == 61u) {
            len = ((lo >> 8) & 0xFFu) + 1u;
            hdr = 2u;
        } else if (len > 61u) {
            special = true;
            hdr = 1u + (len - 60u);
        }
    } else if (kind == 1u) {
        len = 4u + ((tag >> 2) & 7u);
        off = ((tag >> 5) << 8) | ((lo >> 8) & 0xFFu);
        hdr = 2u;
    } else if (kind == 2u) {
        len = (tag >> 2) + 1u;
        off = (lo >> 8) & 0xFFFFu;
        hdr = 3u;
    } else {
        len = (tag >> 2) + 1u;
        off = (lo >> 8) | (hi << 24);
        hdr = 5u;
    }
    const unsigned tokbytes = hdr + (kind == 0u ? len : 0u);
    const bool stopper = special || x >= to || tokbytes > to - x;
    const unsigned nxt = stopper ? 64u : min(lane + tokbytes, 64u);
    while (e < 64u && ws + e < to) {
        unsigned long long T = window_chain(stopper, nxt, lane, e);
        if (T == 0ull) {
            // a literal with 2..4 length bytes: the wavefront moves it, 64 bytes a turn, straight from memory
            unsigned h = 0, llen = 0;
            if (!scan_long_literal((unsigned)__builtin_amdgcn_readlane((int)lo, (int)e), (unsigned)__builtin_amdgcn_readlane((int)hi, (int)e),
                                   ws + e, to, &h, &llen) || op > out_len || llen > out_len - op)
                return false;
            const unsigned at = ws + e + h;
            for (unsigned done = lane; done < llen; done += 64u) {
                L.ptr[op + done] = (uint16_t)(op + done);
                dst[op + done] = src_al[at + done];
            }
            op += llen;
            e += h + llen;                           // (far beyond this window)
            break;
        }
        bool is_tok = __builtin_amdgcn_inverse_ballot_w64(T);
        const int incl = wave_scan_add(is_tok ? (int)len : 0);
        const unsigned o_t = (unsigned)incl - (is_tok ? len : 0u);
        T &= ballot64(o_t + len <= kBrkOwner);                   // (prefix-closed: o_t is monotone; an element is at most 256 bytes)
        is_tok = __builtin_amdgcn_inverse_ballot_w64(T);
        const unsigned last = 63u - (unsigned)__builtin_clzll(T);
        const unsigned N = (unsigned)__builtin_amdgcn_readlane(incl, (int)last);
        const unsigned adv = last + (unsigned)__builtin_amdgcn_readlane((int)tokbytes, (int)last);
        const bool bad = is_tok && ((kind != 0u && (off == 0u || off > op + o_t)) || len > out_len - op - o_t || o_t > out_len - op);
        if (op > out_len || ballot64(bad) != 0ull)
            return false;                            // (a copy from before the block among them: not this kernel's business)
        for (unsigned k4 = lane * 4u; k4 < N; k4 += 256u)
            *reinterpret_cast<uint32_t *>(owner + k4) = 0u;
        if (is_tok)
            owner[o_t] = (uint8_t)(lane + 1u);
        // element attributes for the byte lanes: a0 = o_t (10 bits) | len (9) << 10 | literal << 19 | offset, saturated
        // at 511, << 20 (it only matters where it is shorter than the copy); a1 = where the bytes come from: the
        // literal's place in the staged bytes, or the copy's source position in the block
        const int a0 = (int)(o_t | (len << 10) | (kind == 0u ? (1u << 19) : 0u) | (min(off, 511u) << 20));
        const int a1 = (int)(kind == 0u ? lane + hdr : op + o_t - off);
        unsigned carry = 0;
        for (unsigned B = 0; B < N; B += 64u) {
            const unsigned bb = B + lane;
            const bool active = bb < N;
            int m = active ? (int)owner[bb] : 0;
            m = wave_scan_max(m);
            m = max(m, (int)carry);
            carry = (unsigned)__builtin_amdgcn_readlane(m, 63);
            const unsigned sl = (unsigned)(m - 1) & 63u;
            const unsigned g0 = (unsigned)lane_gather(a0, sl);
            const unsigned g1 = (unsigned)lane_gather(a1, sl);
            const unsigned rel = bb - (g0 & 0x3FFu);
            const unsigned elen = (g0 >> 10) & 0x1FFu, eoff = g0 >> 20;
            const bool lit = ((g0 >> 19) & 1u) != 0u;
            if (active) {
                const unsigned o = op + bb;
                if (lit) {
                    L.ptr[o] = (uint16_t)o;
                    dst[o] = stage[(g1 + rel) & (kBrkStage - 1u)];          // (63 + 2 + 256 bytes from ws at most)
                } else {
                    unsigned r = rel;
                    if (eoff < elen) {               // overlapping copy: every byte names the pattern's first period
                        const unsigned mm = (unsigned)(65536.0f * __builtin_amdgcn_rcpf((float)eoff)) + 1u;
                        r = rel - __umul24(__umul24(rel, mm) >> 16, eoff);
                    }
                    const unsigned src_pos = g1 + r;
                    const unsigned there = L.ptr[src_pos];
                    L.ptr[o] = (uint16_t)(there != 0xFFFFu ? there : src_pos);
                }
            }
        }
        op += N;
        e = adv;
    }
    *exit_p = ws + e;
    *exit_op = op;
    return true;
}

// One block: units[unit_index].  only_full: the first sweep of a launch takes the whole 64 KiB blocks, the second the short
// ones at the ends of their streams -- the last workgroups to finish then finish soon.
__device__ __forceinline__ void brk_block(BrkLds &L, HapGpuDecodeUnit *units, unsigned unit_index, bool only_full, const HapGpuDecodeJob *jobs,
                                          const unsigned long long *__restrict__ recs, const uint4 *__restrict__ joins,
                                          uint32_t *resolved_counter)
{
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const HapGpuDecodeUnit u = units[unit_index];
    if (u.kind != HAPGPU_UNIT_SNAPPY_BLOCK || (u.reserved & HAPGPU_BLOCK_FINE) != 0ull || jobs[u.job].status != 0u ||
        (u.dst_len == kBlockOut) != only_full)
        return;
    // (the same decision the wavefront-per-block kernel takes: 64 KiB blocks run when every 64 KiB mark was found and
    // the stream's 8 KiB pieces, if it has any, did not all check out)
    const HapGpuScanChunk *scan = (const HapGpuScanChunk *)u.aux;
    const bool fine_on = scan->expected_fine != 0u;
    const bool fine_ok = fine_on && scan->found_fine == scan->expected_fine &&
                         __hip_atomic_load(&scan->fine_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    if (scan->ok == 0u || fine_ok || scan->found != scan->expected)
        return;
    const uint32_t *bpos = (const uint32_t *)scan->bpos;
    const unsigned b = (unsigned)u.reserved;
    const unsigned marks = fine_on ? scan->expected_fine : scan->expected;
    const unsigned first = fine_on ? 8u * b : b, last_mark = fine_on ? min(8u * b + 8u, marks) : b + 1u;
    if (!(first < marks && last_mark <= marks))
        return;
    const unsigned from = bpos[first], to = bpos[last_mark], stream_end = bpos[marks];
    const unsigned out_len = u.dst_len;
    if (from >= to || to > stream_end || out_len == 0u || out_len > kBlockOut)
        return;
    const unsigned w0 = from >> 6, nw = ((to - 1u) >> 6) - w0 + 1u;
#ifdef BRK_TIMING
#define BRK_WHY(k) do { if (tid == 0 && resolved_counter) atomicAdd(resolved_counter + 16u + (k), 1u); } while (0)
#else
#define BRK_WHY(k)
#endif
    if (nw > kBrkMaxWindows) {
        BRK_WHY(0);
        return;
    }
    const uint8_t *src_al = (const uint8_t *)u.src;
    uint8_t *dst = (uint8_t *)u.dst;
    const unsigned blk_op = b * kBlockOut;
#ifdef BRK_TIMING
    unsigned long long tstamp[8];
    unsigned rounds_done = 0;
    tstamp[0] = wall_clock64();
#define BRK_STAMP(k) tstamp[k] = wall_clock64()
#else
#define BRK_STAMP(k)
#endif

    // ---- the records of the block's windows: where the element chain enters each, and with how much output behind it ----
    const unsigned seg_bytes = scan_segment_bytes(*scan);
    for (unsigned i = tid; i < nw; i += kBrkThreads) {
        const unsigned ws = (w0 + i) << 6;
        const unsigned seg = ws / seg_bytes, k = (ws % seg_bytes) >> 6;
        unsigned e = 0xFFu, op = 0;
        if (i == 0u) {
            e = from & 63u;                          // the block's mark: an element begins there, with nothing of the block behind it
        } else if (seg < scan->seg_count) {
            const unsigned long long rec = recs[(size_t)(scan->seg_first + seg) * 64u + k];
            const uint4 join = joins[scan->seg_first + seg];
            const unsigned abs_op = join.y + (unsigned)(rec >> 8);
            if (join.x < 64u && k >= join.x && ((unsigned)rec & 0xFFu) < 64u && abs_op - blk_op <= out_len && ws + ((unsigned)rec & 0xFFu) < to) {
                e = (unsigned)rec & 0xFFu;
                op = abs_op - blk_op;
            }
        }
        L.w_entry[i] = (uint8_t)e;
        L.w_op[i] = op;
        L.w_in[i] = 0u;
    }
    // every pointer "not written yet": a copy byte whose source already has its pointer takes that one instead of the
    // source's position -- the windows are worked through roughly in order, so most chains are short before the
    // jumping starts (6.3 rounds without this).  (0xFFFF is no pointer a source can hold: a pointer is below its byte.)
    for (unsigned i = tid * 8u; i < kBlockOut; i += kBrkThreads * 8u)
        *reinterpret_cast<uint4 *>(&L.ptr[i]) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (tid == 0) {
        L.fail = 0u;
        L.ends = 0u;
    }
    __syncthreads();
    BRK_STAMP(1);

    // ---- A. pointers for every output byte of the windows on record (literal bytes to memory); where their chains leave ----
    // (a wavefront's next window is on its way from memory while it works on this one)
    {
        unsigned i = wave;
        uint2 ahead = make_uint2(0u, 0u);
        if (i < nw)
            ahead = brk_load8(src_al, ((w0 + i) << 6) + lane * 8u, stream_end);
        for (; i < nw; i += kBrkWaves) {
            const uint2 bytes = ahead;
            if (i + kBrkWaves < nw)
                ahead = brk_load8(src_al, ((w0 + i + kBrkWaves) << 6) + lane * 8u, stream_end);
            const unsigned e = L.w_entry[i];
            if (e == 0xFFu)
                continue;
            unsigned xp = 0, xo = 0;
            if (!brk_do_window(L, wave, lane, src_al, dst, (w0 + i) << 6, e, L.w_op[i], to, out_len, bytes, &xp, &xo)) {
                L.fail = 1u;
                break;
            }
            if (lane == 0) {
                L.w_exit_p[i] = xp;
                L.w_exit_op[i] = xo;
            }
        }
    }
    __syncthreads();
    if (L.fail != 0u) {
        BRK_WHY(1);
        return;
    }
    BRK_STAMP(2);

    // ---- B. verify the chain; walk what is not on record ----
    if (wave == 0u) {
        // (one wavefront: a window per lane finds out whether its exit is on record; the gaps are walked one after the
        // other -- a handful of windows in one block of twenty)
        for (unsigned base = 0; base < nw; base += 64u) {
            const unsigned i = base + lane;
            bool gap = false;
            unsigned xp = 0, xo = 0;
            if (i < nw && L.w_entry[i] != 0xFFu) {
                xp = L.w_exit_p[i];
                xo = L.w_exit_op[i];
                if (xp < to) {
                    const unsigned j = (xp >> 6) - w0;
                    gap = !(j < nw && L.w_entry[j] == (xp & 63u) && L.w_op[j] == xo);
                }
            }
            unsigned long long gaps = ballot64(gap);
            while (gaps != 0ull) {
                const unsigned g = (unsigned)__builtin_ctzll(gaps);
                gaps &= gaps - 1ull;
                unsigned p = (unsigned)__builtin_amdgcn_readlane((int)xp, (int)g), op = (unsigned)__builtin_amdgcn_readlane((int)xo, (int)g);
                for (unsigned guard = 0; guard <= nw && p < to; guard++) {
                    const unsigned j = (p >> 6) - w0;
                    if (j >= nw) {
                        L.fail = 1u;
                        break;
                    }
                    if (L.w_entry[j] == (p & 63u) && L.w_op[j] == op)
                        break;                       // on record from here on
                    unsigned np = 0, nop = 0;
                    const unsigned ws = (w0 + j) << 6;
                    if (!brk_do_window(L, 0u, lane, src_al, dst, ws, p & 63u, op, to, out_len, brk_load8(src_al, ws + lane * 8u, stream_end), &np, &nop)) {
                        L.fail = 1u;
                        break;
                    }
                    if (lane == 0) {
                        L.w_entry[j] = (uint8_t)(p & 63u);
                        L.w_op[j] = op;
                        L.w_exit_p[j] = np;
                        L.w_exit_op[j] = nop;
                    }
                    p = np;
                    op = nop;
                }
            }
        }
    }
    __syncthreads();
    if (L.fail != 0u) {
        BRK_WHY(2);
        return;
    }
    // every window on the chain is entered exactly once (the first: by nobody), and one chain ends the block
    for (unsigned i = tid; i < nw; i += kBrkThreads) {
        if (L.w_entry[i] == 0xFFu)
            continue;
        const unsigned xp = L.w_exit_p[i], xo = L.w_exit_op[i];
        if (xp == to && xo == out_len) {
            atomicAdd(&L.ends, 1u);
        } else if (xp < to && (xp >> 6) - w0 < nw && (xp >> 6) - w0 > i && xo <= out_len) {
            const unsigned j = (xp >> 6) - w0;
            if (L.w_entry[j] == (xp & 63u) && L.w_op[j] == xo) {
                // (byte counters: four windows share a word)
                atomicAdd(reinterpret_cast<uint32_t *>(L.w_in) + (j >> 2), 1u << (8u * (j & 3u)));
            } else {
                L.fail = 1u;
            }
        } else {
            L.fail = 1u;
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < nw; i += kBrkThreads)
        if (L.w_entry[i] != 0xFFu && L.w_in[i] != (i == 0u ? 0u : 1u))
            L.fail = 1u;
    __syncthreads();
    if (L.fail != 0u || L.ends != 1u || L.w_entry[0] == 0xFFu) {
        BRK_WHY(L.fail != 0u ? 3 : L.ends != 1u ? 4 : 5);
        return;
    }
    BRK_STAMP(3);

    // ---- C. pointer jumping: a group of four whose pointers have stopped moving is left alone from then on ----
    {
        unsigned pending = 0xFFFFu;                  // bit it: the thread's group (it * 1024 + tid) * 4 is not final yet
        unsigned round = 0;
        for (; round < kBrkMaxRounds; round++) {
            int changed = 0;
#pragma unroll 1
            for (unsigned it = 0; it < 16u; it++) {
                if (!((pending >> it) & 1u))
                    continue;
                const unsigned base = (it * kBrkThreads + tid) * 4u;
                if (base >= out_len) {
                    pending &= ~(1u << it);
                    continue;
                }
                const uint2 v = *reinterpret_cast<const uint2 *>(&L.ptr[base]);
                unsigned p[4] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16};
                const unsigned n = min(4u, out_len - base);
#pragma unroll
                for (unsigned k = 0; k < 4u; k++)
                    if (k >= n)
                        p[k] = base + k;             // (beyond a short block's end: nobody's bytes)
                unsigned q[4];
#pragma unroll
                for (unsigned k = 0; k < 4u; k++)
                    q[k] = L.ptr[p[k]];
                if (q[0] == p[0] && q[1] == p[1] && q[2] == p[2] && q[3] == p[3]) {
                    pending &= ~(1u << it);          // (what they point at points at itself: literal bytes)
                } else {
                    *reinterpret_cast<uint2 *>(&L.ptr[base]) = make_uint2(q[0] | (q[1] << 16), q[2] | (q[3] << 16));
                    changed = 1;
                }
            }
            if (!__syncthreads_or(changed))
                break;
        }
        if (round == kBrkMaxRounds)
            return;                                  // (never seen: 2^18 links; the other kernel writes the block again)
#ifdef BRK_TIMING
        rounds_done = round;
#endif
    }
    BRK_STAMP(4);
    // ---- D. every copy byte fetches the literal byte its pointer names ----
    // (Ordinary cached loads: the literal bytes were written -- by whichever wavefront -- before the fence and the barrier
    // below, this kernel has not read a byte of the block until now, and a CU's L1 starts a kernel empty; the copy bytes
    // stored meanwhile may or may not show in a line that is already here, and nobody reads THEM.  Loads past the L1,
    // one request per lane, took 72 us a block; sixteen loads of a lane are in flight together.)
    // (a fence of WORKGROUP scope: the readers sit on this CU.  One of agent scope writes the XCD's L2 back and
    // invalidates it -- 26 us a block)
    __threadfence_block();
    __syncthreads();
#pragma unroll 1
    for (unsigned base = tid; base < out_len; base += kBrkThreads * 16u) {
        unsigned r[16], v[16];
#pragma unroll
        for (unsigned k = 0; k < 16u; k++) {
            const unsigned o = base + k * kBrkThreads;
            r[k] = o < out_len ? L.ptr[o] : o;
        }
#pragma unroll
        for (unsigned k = 0; k < 16u; k++) {
            const unsigned o = base + k * kBrkThreads;
            v[k] = 0u;
            if (r[k] != o)
                v[k] = dst[r[k]];
        }
#pragma unroll
        for (unsigned k = 0; k < 16u; k++) {
            const unsigned o = base + k * kBrkThreads;
            if (r[k] != o)
                dst[o] = (uint8_t)v[k];
        }
    }
    BRK_STAMP(5);
    if (tid == 0) {
        units[unit_index].kind = HAPGPU_UNIT_SKIP;
        if (resolved_counter)
            atomicAdd(resolved_counter, 1u);
#ifdef BRK_TIMING
        if (resolved_counter) {
            for (unsigned k = 0; k < 5u; k++)
                atomicAdd(resolved_counter + 2u + k, (unsigned)(tstamp[k + 1u] - tstamp[k]));
            atomicAdd(resolved_counter + 8u, rounds_done);
            atomicAdd(resolved_counter + 9u, nw);
        }
#endif
    }
}

// As many workgroups as the GPU has CUs (each fills one: 158 KiB of LDS), every one taking blocks off a counter until there
// are none: block after block without a dispatch in between (a grid of one workgroup per block took 350 us for the 528
// blocks of an 8K frame, this takes 258), full blocks first.  The blocks are those of the call's scanned streams (scan_merge
// wrote their units behind each stream's own): stream c's block b is units[chunks[c].unit + 1 + b].
constexpr unsigned kBrkMaxStreams = 256u;
__global__ __launch_bounds__(kBrkThreads) void snappy_decode_block_resolve_kernel(HapGpuDecodeUnit *units, unsigned unit_count,
                                                                                  const HapGpuScanChunk *chunks, unsigned chunk_count,
                                                                                  const HapGpuDecodeJob *jobs,
                                                                                  const unsigned long long *__restrict__ recs,
                                                                                  const uint4 *__restrict__ joins,
                                                                                  uint32_t *resolved_counter, uint32_t *work_counter)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dynamic_lds[];
    BrkLds &L = *reinterpret_cast<BrkLds *>(dynamic_lds);
    const unsigned tid = threadIdx.x;
    if (chunk_count == 0u || chunk_count > kBrkMaxStreams)
        return;
    if (tid < chunk_count) {
        const HapGpuScanChunk sc = chunks[tid];
        // (what scan_merge wrote: `expected` blocks behind the stream's unit, inside the slots the host reserved; none for a
        // stream whose 64 KiB blocks do not run at all: marks not all found, or -- this library's own plain frames -- decoded
        // by its 8 KiB pieces)
        const bool pieces = sc.expected_fine != 0u && sc.found_fine == sc.expected_fine && sc.fine_failed == 0u;
        L.stream_first[tid + 1u] = (sc.ok != 0u && !pieces && sc.found == sc.expected && sc.expected <= sc.slots &&
                                    (unsigned long long)sc.unit + 1u + sc.expected <= unit_count) ? sc.expected : 0u;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        L.stream_first[0] = 0u;
        for (unsigned c = 1; c <= chunk_count; c++) {
            run += L.stream_first[c];
            L.stream_first[c] = run;
        }
    }
    __syncthreads();
    const unsigned total = L.stream_first[chunk_count];
    for (;;) {
        if (tid == 0)
            L.work = atomicAdd(work_counter, 1u);
        __syncthreads();
        const unsigned idx = L.work;
        __syncthreads();
        if (idx >= 2u * total)
            break;
        const bool only_full = idx < total;
        const unsigned j = only_full ? idx : idx - total;
        unsigned lo = 0, hi = chunk_count;          // stream c: stream_first[c] <= j < stream_first[c + 1]
        while (hi - lo > 1u) {
            const unsigned mid = (lo + hi) >> 1;
            if (L.stream_first[mid] <= j)
                lo = mid;
            else
                hi = mid;
        }
        brk_block(L, units, chunks[lo].unit + 1u + (j - L.stream_first[lo]), only_full, jobs, recs, joins, resolved_counter);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// group tables for the 8 KiB pieces of this library's table-less streams, from the block scan's records
// ------------------------------------------------------------------------------------------
//
// A plain hap.h frame of this library (hap.c:448-476's layout, no private section) is decoded fastest by the block-per-lane
// kernel of snappy_decode_fields.hip, which wants 64 starting points inside every 8 KiB fragment.  Until round 5 a lane
// per fragment walked the fragment's ~400 elements twice to find them (guess_group_tables_kernel: 1.95 ms per 60 8K
// frames, bound by scattered loads).  But the block scan has already walked every element of the stream: its records say,
// for every window of 64 compressed bytes, at which byte the element chain enters it, how many output bytes and -- since
// round 6 -- how many ELEMENTS the chain has behind it there.  So a wavefront per fragment now: the fragment's bytes staged
// in LDS, the records of its windows in a list; lane 0 counts the elements of the first window that lie in front of the
// fragment, lane 1 those of the last window up to the fragment's end (N = the difference); then lane g looks up the window
// that holds element g * ceil(N / 64) and walks the few elements from that window's entry to it.  A dozen element steps per
// lane instead of eight hundred.  The table is a HINT, as every table is: the block-per-lane kernel verifies all it says
// while it decodes, and a piece this kernel declines (a window without a record of the true chain, an element that is no
// field-stream element) stays a unit of the other kernel.
constexpr unsigned kGtMaxCompressed = HAPGPU_SCAN_FINE + 320u;                  // what a field-stream fragment compresses to at most
constexpr unsigned kGtMaxWindows = (kGtMaxCompressed + 63u) / 64u + 2u;
constexpr unsigned kGtStage = kGtMaxWindows * 64u + 16u;
constexpr unsigned kGtBatchPieces = 16384u;         // calls with room for that many pieces and more (four 8K frames) decline pieces with unrecorded windows
constexpr unsigned kGtMaxSteps = 384u;              // elements a lane walks at most: a window holds up to 32, a dozen on average

// the element at byte x of the staged fragment: output bytes and stream bytes.  CHECK: is it one a field stream has -- the
// promises of the fragment table: it begins at output position p of the fragment on a field boundary (`starts`: a bit per
// byte of a block), stays inside its 128-byte half-tile, a copy comes a whole number of blocks from inside the fragment?
// The walks that only look for a place do not ask (they pass over elements twice and over some in front of the fragment); the
// walk over a lane's own group does, and between them the groups are the whole fragment.
template <bool CHECK>
__device__ __forceinline__ bool gt_element(const uint8_t *stage, unsigned x, unsigned p, unsigned block, unsigned starts, unsigned *len, unsigned *adv)
{
    const uint32_t *st32 = reinterpret_cast<const uint32_t *>(stage);
    const unsigned w = __builtin_amdgcn_alignbyte(st32[(x >> 2) + 1u], st32[x >> 2], x);
    const unsigned kind = w & 3u, up = (w >> 2) & 63u;
    bool ok;
    if (kind == 0u) {
        *len = (up == 60u ? ((w >> 8) & 255u) : up) + 1u;
        *adv = *len + (up == 60u ? 2u : 1u);
        ok = up <= 60u;
    } else {
        *len = kind == 1u ? ((w >> 2) & 7u) + 4u : up + 1u;
        *adv = kind + 1u;
        ok = kind != 3u;
        if (CHECK) {
            const unsigned off = kind == 1u ? (((w >> 5) & 7u) << 8) | ((w >> 8) & 255u) : (w >> 8) & 0xFFFFu;
            ok = ok && off >= block && (off & (block - 1u)) == 0u && off <= p;                       // (blocks are 8 or 16 bytes)
        }
    }
    if (CHECK)
        ok = ok && ((starts >> (p & (block - 1u))) & 1u) != 0u && (p & 127u) + *len <= 128u;
    return ok;
}

__global__ __launch_bounds__(64) void group_tables_from_records_kernel(HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                                                       const uint32_t *__restrict__ work,
                                                                       const unsigned long long *__restrict__ recs,
                                                                       const uint4 *__restrict__ joins)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[kGtStage];
    __shared__ uint32_t w_pos[kGtMaxWindows + 1u], w_e[kGtMaxWindows + 1u], w_op[kGtMaxWindows + 1u];
    __shared__ uint32_t g_c[65], g_o[65];
    const unsigned lane = threadIdx.x;
    if (blockIdx.x >= work[0])
        return;
    const unsigned idx = work[1u + blockIdx.x];
    if (idx >= unit_count)
        return;
    HapGpuDecodeUnit u = units[idx];
    if (u.kind != HAPGPU_UNIT_SNAPPY_BLOCK || !(u.reserved & HAPGPU_BLOCK_FINE) || u.aux == 0u)
        return;
    // (the job's and the stream's words in one round of loads: a wavefront of this kernel is a chain of trips to memory)
    const HapGpuDecodeJob *job = &jobs[u.job];
    const HapGpuScanChunk *scan = (const HapGpuScanChunk *)u.aux;
    const unsigned job_flags = job->reserved, job_status = job->status, layout = job->fields_period;
    const uint64_t job_tables = job->group_tables;
    const unsigned marks = scan->expected_fine, scan_ok = scan->ok, scan_found = scan->found_fine;
    const uint32_t *bpos = (const uint32_t *)scan->bpos;
    if (!((job_flags >> 16) & 1u) || job_tables == 0u || job_status != 0u)
        return;
    const unsigned block = (layout == 4u || layout == 8u) ? 16u : 8u;
    // field starts inside a block, a bit per byte: [2,6,4,4]: 0, 2, 8, 12; [4,4]: 0, 4; [2,6]: 0, 2; [4,4,4,4]: 0, 4, 8, 12
    const unsigned starts = layout == 4u ? 0x1105u : layout == 2u ? 0x11u : layout == 6u ? 0x05u : 0x1111u;
    const unsigned b = (unsigned)u.reserved;
    if (!scan_ok || marks == 0u || scan_found != marks || b + 1u > marks)
        return;
    const unsigned from = bpos[b], to = bpos[b + 1u], stream_end = bpos[marks];
    const unsigned out_len = u.dst_len;
    if (from >= to || to > stream_end || to - from > kGtMaxCompressed || out_len == 0u || out_len > HAPGPU_SCAN_FINE || (out_len % block) != 0u)
        return;
    const unsigned n = to - from;
    const unsigned w0 = from >> 6, nw = ((to - 1u) >> 6) - w0 + 1u;          // <= kGtMaxWindows - 1
    const uint8_t *src_al = (const uint8_t *)u.src;
    const unsigned base = w0 << 6;                                            // stream coordinate of stage[0]
    const unsigned seg_bytes = scan_segment_bytes(*scan);
    const unsigned blk_op = b * HAPGPU_SCAN_FINE;

    // the records of the fragment's first 64 windows are asked for in front of its bytes: both are on their way together
    const unsigned seg_first = scan->seg_first, seg_count = scan->seg_count;
    unsigned long long rec0 = kRecNone;
    uint4 join0 = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
    if (lane < nw) {
        const unsigned ws = base + (lane << 6);
        const unsigned seg = ws / seg_bytes;
        if (seg < seg_count) {
            rec0 = recs[(size_t)(seg_first + seg) * 64u + ((ws % seg_bytes) >> 6)];
            join0 = joins[seg_first + seg];
        }
    }
    // the fragment's bytes (and eight more: an element's tag is read as a dword)
    for (unsigned c = lane * 16u; c < nw * 64u + 16u; c += 1024u)
        *reinterpret_cast<uint4 *>(stage + c) = scan_load16(src_al, base + c, stream_end);
    // the windows on record, in order: where the chain enters, its element number and output position there
    unsigned m = 0;
    bool gap = false;
    for (unsigned i0 = 0; i0 < nw; i0 += 64u) {
        const unsigned i = i0 + lane;
        bool usable = false;
        unsigned pos = 0, e = 0, op = 0;
        if (i < nw) {
            const unsigned ws = base + (i << 6);
            const unsigned seg = ws / seg_bytes, k = (ws % seg_bytes) >> 6;
            if (seg < seg_count) {
                const unsigned long long rec = i0 == 0u ? rec0 : recs[(size_t)(seg_first + seg) * 64u + k];
                const uint4 join = i0 == 0u ? join0 : joins[seg_first + seg];
                const unsigned entry = (unsigned)rec & 0xFFu;
                if (join.x < 64u && k >= join.x && entry < 64u && ws + entry < to) {
                    usable = true;
                    pos = ws + entry;
                    e = join.z + (unsigned)(rec >> 40);
                    op = join.y + (unsigned)(rec >> 8) - blk_op;
                } else if (join.x >= 64u || (join.w != 0u && k < join.x)) {
                    // A window the true chain may enter without a record of its own (the first windows of a segment whose
                    // guessed chain joined late) is not on the list: the walks below start at the last listed window in
                    // front of what they look for and pass through it -- a lane then walks dozens of elements while 63
                    // wait.  In a call of few pieces that beats leaving the piece to the other kernel (one plain 8K frame:
                    // its generic launch 90 us -> 7); in a batch it does not (60 frames: this kernel 0.78 -> 1.14 ms for
                    // 0.09 ms less of the other), and the piece is declined.
                    gap = true;
                }
            } else {
                gap = true;
            }
        }
        const unsigned long long mask = ballot64(usable);
        if (usable) {
            const unsigned at = m + (unsigned)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            w_pos[at] = pos;
            w_e[at] = e;
            w_op[at] = op;
        }
        m += (unsigned)__builtin_popcountll(mask);
    }
    if (m == 0u || (gridDim.x >= kGtBatchPieces && ballot64(gap) != 0ull))
        return;
    __syncthreads();
    if (w_pos[0] > from)
        return;                                      // (the fragment begins with an element: its window has a record at or in front of it)
    // ---- N: lane 0 counts up to the fragment's first element, lane 1 from the last window on record to its end ----
    unsigned cnt = 0, ok = 1u;
    {
        unsigned pos = lane == 0u ? w_pos[0] : w_pos[m - 1u], op = lane == 0u ? w_op[0] : w_op[m - 1u];
        const unsigned stop = lane == 0u ? from : to;
        for (unsigned step = 0; step < kGtMaxSteps && lane < 2u && pos < stop; step++) {
            unsigned len, adv;
            if (!gt_element<false>(stage, pos - base, op, block, starts, &len, &adv))
                ok = 0u;
            pos += adv;
            op += len;
            cnt += 1u;
        }
        if (lane < 2u && (pos != stop || op != (lane == 0u ? 0u : out_len)))
            ok = 0u;
    }
    const unsigned e0 = w_e[0] + (unsigned)__builtin_amdgcn_readlane((int)cnt, 0);
    const unsigned e1 = w_e[m - 1u] + (unsigned)__builtin_amdgcn_readlane((int)cnt, 1);
    const unsigned N = e1 - e0;
    if (ballot64(ok == 0u) != 0ull || e1 <= e0 || 4u * N > out_len)
        return;
    const unsigned G = (N + 63u) >> 6;
    // ---- lane g: where element g * G begins ----
    unsigned cpos = n, opos = out_len, end_c = n, end_o = out_len;
    if (lane * G < N) {
        const unsigned T = e0 + lane * G;
        unsigned lo = 0, hi = m;                     // the last window on record with w_e <= T (w_e[0] <= e0 <= T)
        while (hi - lo > 1u) {
            const unsigned mid = (lo + hi) >> 1;
            if (w_e[mid] <= T)
                lo = mid;
            else
                hi = mid;
        }
        unsigned pos = w_pos[lo], e = w_e[lo], op = w_op[lo];
        for (unsigned step = 0; step < kGtMaxSteps && e < T && pos < to; step++) {
            unsigned len, adv;
            if (!gt_element<false>(stage, pos - base, op, block, starts, &len, &adv))
                ok = 0u;
            pos += adv;
            op += len;
            e += 1u;
        }
        if (e != T || pos < from || pos > to || op > out_len)
            ok = 0u;
        cpos = pos - from;
        opos = op;
        // ... and on through the group's own G elements: between them the lanes pass over every element of the fragment, so
        // a piece that is no field stream (another compressor's bytes with marks in the right places) is declined HERE, for
        // the price of a few steps -- not by the block-per-lane kernel, whose verdict sends the whole frame round again
        const unsigned T1 = min(T + G, e1);
        for (unsigned step = 0; step < G && e < T1 && pos < to; step++) {
            unsigned len, adv;
            if (!gt_element<true>(stage, pos - base, op, block, starts, &len, &adv))
                ok = 0u;
            pos += adv;
            op += len;
            e += 1u;
        }
        if (e != T1 || pos > to || op > out_len)
            ok = 0u;
        end_c = pos - from;
        end_o = op;
    }
    g_c[lane] = cpos;
    g_o[lane] = opos;
    if (lane == 0) {
        g_c[64] = n;
        g_o[64] = out_len;
    }
    __syncthreads();
    const unsigned gsz = g_c[lane + 1u] - cpos, gout = g_o[lane + 1u] - opos;
    if (g_c[lane + 1u] != end_c || g_o[lane + 1u] != end_o || end_c < cpos || end_o < opos || gsz >= 4096u || gout >= 4096u)
        ok = 0u;                                     // (a group ends where the next one begins)
    if (ballot64(ok == 0u) != 0ull)
        return;
    // fragment table version 4: 64 x 24 bits (compressed bytes | bytes produced << 12), then the element count
    uint8_t *table = (uint8_t *)(uintptr_t)(job_tables + (uint64_t)idx * HAP_GROUP_TABLE_BYTES);
    const unsigned entry = gsz | (gout << 12);
    table[3u * lane] = (uint8_t)entry;
    table[3u * lane + 1u] = (uint8_t)(entry >> 8);
    table[3u * lane + 2u] = (uint8_t)(entry >> 16);
    if (lane == 0) {
        *reinterpret_cast<uint32_t *>(table + 192) = N;          // (bytes 192, 193: the count; 194, 195: zero.  The arena and 196 are multiples of 4)
        const uint64_t begin = (uint64_t)(uintptr_t)src_al + from, end = begin + n, section_end = job->payload + job->payload_len;
        u.src = begin;
        u.src_len = n;
        u.kind = layout == 4u ? HAPGPU_UNIT_SNAPPY_FIELDS4 : layout == 2u ? HAPGPU_UNIT_SNAPPY_FIELDS2
               : layout == 8u ? HAPGPU_UNIT_SNAPPY_FIELDS44 : HAPGPU_UNIT_SNAPPY_FIELDS26;
        u.aux = (uint64_t)(uintptr_t)table;
        u.reserved = section_end > end ? (section_end - end < 15u ? section_end - end : 15u) : 0u;
        units[idx] = u;
    }
}

} // namespace

#ifdef BRK_TIMING
extern "C" void hapgpu_debug_merge_counters(unsigned *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_merge_dbg), sizeof(unsigned) * 8u);
}
#endif

extern "C" int hapgpu_launch_decode_plan(HapGpuDecodeJob *jobs, unsigned job_count, unsigned max_chunks, hipStream_t stream)
{
    if (job_count == 0)
        return 0;
    hipLaunchKernelGGL(decode_plan_kernel, dim3(job_count), dim3(64), 0, stream, jobs, job_count);
    if (max_chunks)
        hipLaunchKernelGGL(decode_expand_kernel, dim3(max_chunks, job_count), dim3(64), 0, stream, jobs, job_count);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// frag_log2: fragment size of FRAGMENT units in this batch (0 = none present).
// dynamic LDS to request for the v2 kernel: none when its ring + tail fit the static array
static constexpr unsigned fragment_dynamic_lds(unsigned ring) { return ring + kFragmentTail <= 65536u ? 0u : ring + kFragmentTail; }

extern "C" int hapgpu_launch_snappy_decode_fields(const HapGpuDecodeUnit *units, unsigned unit_count, HapGpuDecodeJob *jobs,
                                                  unsigned fields_kinds, hipStream_t stream);

// The pieces the block scan listed in `work` ([0]: how many; room for work_slots): their group tables from the scan's records.
extern "C" int hapgpu_launch_group_tables_from_records(HapGpuDecodeUnit *units, unsigned unit_count, const HapGpuDecodeJob *jobs,
                                                       const uint32_t *work, unsigned work_slots, const void *recs, const void *joins,
                                                       hipStream_t stream)
{
    if (unit_count == 0 || work_slots == 0 || !work || !recs || !joins)
        return 0;
    hipLaunchKernelGGL(group_tables_from_records_kernel, dim3(work_slots), dim3(64), 0, stream, units, unit_count, jobs, work,
                       (const unsigned long long *)recs, (const uint4 *)joins);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}



// Finds the 64 KiB blocks of the whole-stream units listed in `chunks` (see the block scan above) and writes their
// BLOCK units; the decode launch that follows must include the stream kernel.
extern "C" int hapgpu_launch_scan_blocks(HapGpuDecodeUnit *units, const HapGpuDecodeJob *jobs, HapGpuScanChunk *chunks,
                                         unsigned chunk_count, HapGpuScanSegment *segs, void *recs, void *joins,
                                         unsigned seg_total, uint32_t *fine_work, unsigned fine_first, unsigned fine_pool,
                                         hipStream_t stream)
{
    if (chunk_count == 0 || seg_total == 0)
        return 0;
    // (a single 8K frame is ~3 000 segments; the GPU holds 8 192 wavefronts)
    const unsigned warm = seg_total <= 8192u ? kScanWarmWindowsShortCall : kScanWarmWindows;
    hipLaunchKernelGGL(scan_walk_kernel, dim3(seg_total), dim3(64), kScanLdsFor(warm), stream, units, jobs, chunks, chunk_count, segs,
                       (unsigned long long *)recs, (uint4 *)joins, seg_total, warm);
    // (fine_work: [0] the list's length, then the list -- fine_pool entries --, then the pool's cursor)
    hipLaunchKernelGGL(scan_merge_kernel, dim3(chunk_count), dim3(64), 0, stream, units, jobs, chunks, chunk_count, segs,
                       (const unsigned long long *)recs, (uint4 *)joins, fine_work ? fine_work + 1u + fine_pool : nullptr, fine_first, fine_pool);
    hipLaunchKernelGGL(scan_find_kernel, dim3(seg_total), dim3(64), 0, stream, units, chunks, chunk_count, segs,
                       (const unsigned long long *)recs, (const uint4 *)joins, seg_total, 0u);
    hipLaunchKernelGGL(scan_find_kernel, dim3(seg_total), dim3(64), 0, stream, units, chunks, chunk_count, segs,
                       (const unsigned long long *)recs, (const uint4 *)joins, seg_total, 1u);
    if (fine_work)
        hipLaunchKernelGGL(scan_decide_kernel, dim3(chunk_count), dim3(64), 0, stream, chunks, chunk_count, fine_work);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// scan_recs / scan_joins: what the block scan of THIS call left (hapgpu_launch_scan_blocks), or null; scan_blocks_hint: about
// how many 64 KiB blocks its streams hold: few enough, and a workgroup each takes them first
// (snappy_decode_block_resolve_kernel); resolved: a counter of the blocks that went through
extern "C" int hapgpu_launch_snappy_decode(const HapGpuDecodeUnit *units, unsigned unit_count, HapGpuDecodeJob *jobs,
                                           unsigned frag_log2, unsigned fragment_kinds, int any_stream_or_copy_units,
                                           const uint32_t *fine_work, unsigned fine_slots, const void *scan_recs,
                                           const void *scan_joins, const HapGpuScanChunk *scan_chunks, unsigned scan_chunk_count,
                                           unsigned scan_blocks_hint, uint32_t *resolved, hipStream_t stream)
{
    if (unit_count == 0)
        return 0;
    // field streams (fragment table version 3): the block-per-lane decoder of snappy_decode_fields.hip
    // (bit 12: the pre-pass may have turned 8 KiB pieces of scanned streams into such units: they lie behind the ordinary ones)
    if ((fragment_kinds >> 8) & 15u) {
        if (hapgpu_launch_snappy_decode_fields(units, unit_count + (((fragment_kinds >> 12) & 1u) ? fine_slots : 0u), jobs,
                                               (fragment_kinds >> 8) & 15u, stream) != 0)
            return 4;
        fragment_kinds &= 0xFFu;
        if (fragment_kinds == 0u)
            frag_log2 = 0u;
    }
    if (any_stream_or_copy_units) {
        // any_stream_or_copy_units == 2: most units are 64 KiB blocks found by the block scan -- a 2 KiB ring (copies
        // from further back re-read the output from memory) lets 32 wavefronts share a CU instead of 4
        static bool once = false;
        static unsigned ring_forced = 0;
        if (!once) {
            (void)hipFuncSetAttribute((const void *)snappy_decode_fragment_kernel<65536u, true, 1u>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + kFragmentTail);
            const char *e = HAP_AB_ENV("HAP_AMD_STREAM_RING_LOG2");
            if (e && atoi(e) >= 11 && atoi(e) <= 16)
                ring_forced = (unsigned)atoi(e);
            once = true;
        }
        // (3: every stream of the call is as short as one 8 KiB fragment -- fine chunks: thousands of short streams want
        // wavefronts per CU more than they want their whole output in the ring; 30 8K frames: 2.62 ms against 3.53 with an 8 KiB ring)
        const unsigned ring_log2 = ring_forced ? ring_forced : (any_stream_or_copy_units == 2 || any_stream_or_copy_units == 3) ? 11u : 15u;
        // with the block scan: the 8 KiB blocks it listed first (phase 1, over the list's capacity: the count is on the
        // device), then the ordinary units (phase 2) -- a stream whose 8 KiB pieces turned out not to be independent is
        // decoded by its 64 KiB blocks or whole in the second launch
        const bool two = any_stream_or_copy_units == 2 && fine_work != nullptr && fine_slots != 0u;
        // A workgroup per 64 KiB block shortens the CALL -- a block takes 0.1 ms instead of 0.8 -- at about twice the work
        // per block (the block scan's records are verified, the pointers jump) and with one workgroup per CU: it pays while
        // the blocks are few enough for the wavefront-per-block kernel to leave most of the GPU idle.  Measured on an
        // MI355X, decode kernels of a call, 8K frames of the reference encoder (528 blocks each): DXT5 1 frame 0.82 -> 0.26 ms,
        // 2 frames 0.89 -> 0.49, 3 frames 0.95 -> 0.91, 4 frames 1.00 -> 1.21; Hap Q (more elements per block) 1 frame 0.38,
        // 2 frames 0.68, 3 frames 1.11 against 1.12 for FOUR frames the other way: up to four blocks per CU (the host's
        // estimate: what the scanned streams' textures hold).
        static int resolve_on = -1;
        static unsigned resolve_max_units = 0, resolve_workgroups = 256u;
        if (resolve_on < 0) {
            const char *e = HAP_AB_ENV("HAP_AMD_BLOCK_RESOLVE");
            int dev = 0;
            hipDeviceProp_t prop;
            resolve_on = e ? atoi(e) : 1;
            resolve_max_units = 4u * 256u;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) {
                resolve_workgroups = (unsigned)prop.multiProcessorCount;
                resolve_max_units = 4u * resolve_workgroups;
            }
            if (resolve_on > 1)
                resolve_max_units = 0xFFFFFFFFu;         // (measurement builds: every call)
            if (hipFuncSetAttribute((const void *)snappy_decode_block_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sizeof(BrkLds)) != hipSuccess) {
                (void)hipGetLastError();
                resolve_on = 0;
            }
        }
        for (unsigned phase = two ? 1u : any_stream_or_copy_units == 2 ? 2u : 0u; phase <= (any_stream_or_copy_units == 2 ? 2u : 0u); phase++) {
            const dim3 grid(phase == 1u ? fine_slots : unit_count);
            // the 64 KiB blocks of the scanned streams, a workgroup each -- after the 8 KiB pieces of phase 1 (whose failures
            // decide which units run), in front of the wavefront-per-unit launch that takes whatever is left
            if (phase == 2u && resolve_on && scan_recs && scan_joins && scan_chunks && resolved && scan_chunk_count <= kBrkMaxStreams &&
                scan_blocks_hint <= resolve_max_units) {
                // (resolved[0]: blocks that went through, ever; [1]: this launch's work counter)
                (void)hipMemsetAsync(resolved + 1, 0, sizeof(uint32_t), stream);
                hipLaunchKernelGGL(snappy_decode_block_resolve_kernel, dim3(resolve_workgroups), dim3(kBrkThreads), sizeof(BrkLds), stream,
                                   const_cast<HapGpuDecodeUnit *>(units), unit_count, scan_chunks, scan_chunk_count, jobs,
                                   (const unsigned long long *)scan_recs, (const uint4 *)scan_joins, resolved, resolved + 1);
                // (an accelerator only: a launch the runtime refuses -- 159 KiB of LDS is nearly all a CU has -- must not fail
                // the call: the wavefront-per-block launch below decodes every block then, as it did before)
                if (hipGetLastError() != hipSuccess)
                    resolve_on = 0;
            }
            if (ring_log2 == 11)
                hipLaunchKernelGGL((snappy_decode_fragment_kernel<2048u, true, 1u>), grid, dim3(64), 0, stream, units, grid.x, jobs, phase, fine_work);
            else if (ring_log2 == 12)
                hipLaunchKernelGGL((snappy_decode_fragment_kernel<4096u, true, 1u>), grid, dim3(64), 0, stream, units, grid.x, jobs, phase, fine_work);
            else if (ring_log2 == 13)
                hipLaunchKernelGGL((snappy_decode_fragment_kernel<8192u, true, 1u>), grid, dim3(64), 0, stream, units, grid.x, jobs, phase, fine_work);
            else if (ring_log2 == 14)
                hipLaunchKernelGGL((snappy_decode_fragment_kernel<16384u, true, 1u>), grid, dim3(64), fragment_dynamic_lds(16384u), stream, units, grid.x, jobs, phase, fine_work);
            else if (ring_log2 == 15)
                hipLaunchKernelGGL((snappy_decode_fragment_kernel<32768u, true, 1u>), grid, dim3(64), fragment_dynamic_lds(32768u), stream, units, grid.x, jobs, phase, fine_work);
            else
                hipLaunchKernelGGL((snappy_decode_fragment_kernel<65536u, true, 1u>), grid, dim3(64), fragment_dynamic_lds(65536u), stream, units, grid.x, jobs, phase, fine_work);
        }
    }
    static bool once16 = false;
    if (!once16 && frag_log2 == 16) {
        (void)hipFuncSetAttribute((const void *)snappy_decode_fragment_kernel<65536u, false, 1u>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + kFragmentTail);
        (void)hipFuncSetAttribute((const void *)snappy_decode_fragment_kernel<65536u, false, 2u>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + kFragmentTail);
        (void)hipFuncSetAttribute((const void *)snappy_decode_fragment_kernel<65536u, false, 4u>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + kFragmentTail);
        once16 = true;
    }
#define HAP_LAUNCH_FRAGMENT(RINGBYTES)                                                                                          \
    do {                                                                                                                        \
        if (fragment_kinds & 1u)                                                                                                \
            hipLaunchKernelGGL((snappy_decode_fragment_kernel<RINGBYTES, false, 1u>), dim3(unit_count), dim3(64),               \
                               fragment_dynamic_lds(RINGBYTES), stream, units, unit_count, jobs, 0u, nullptr);                               \
        if (fragment_kinds & 2u)                                                                                                \
            hipLaunchKernelGGL((snappy_decode_fragment_kernel<RINGBYTES, false, 2u>), dim3(unit_count), dim3(64),               \
                               fragment_dynamic_lds(RINGBYTES), stream, units, unit_count, jobs, 0u, nullptr);                               \
        if (fragment_kinds & 4u)                                                                                                \
            hipLaunchKernelGGL((snappy_decode_fragment_kernel<RINGBYTES, false, 4u>), dim3(unit_count), dim3(64),               \
                               fragment_dynamic_lds(RINGBYTES), stream, units, unit_count, jobs, 0u, nullptr);                               \
    } while (0)
    // 8 KiB fragments whose table promises a 3 KiB match window: 4 KiB ring, twice the waves per CU
    if (frag_log2 == 13u) {
        if (fragment_kinds & 16u)
            hipLaunchKernelGGL((snappy_decode_fragment_kernel<4096u, false, 1u, true>), dim3(unit_count), dim3(64), 0, stream, units, unit_count, jobs, 0u, nullptr);
        if (fragment_kinds & 32u)
            hipLaunchKernelGGL((snappy_decode_fragment_kernel<4096u, false, 2u, true>), dim3(unit_count), dim3(64), 0, stream, units, unit_count, jobs, 0u, nullptr);
        if (fragment_kinds & 64u)
            hipLaunchKernelGGL((snappy_decode_fragment_kernel<4096u, false, 4u, true>), dim3(unit_count), dim3(64), 0, stream, units, unit_count, jobs, 0u, nullptr);
    }
    switch (frag_log2) {
    case 0: break;
    case 10: case 11: case 12: case 13: HAP_LAUNCH_FRAGMENT(8192u); break;
    case 14: HAP_LAUNCH_FRAGMENT(16384u); break;
    case 15: HAP_LAUNCH_FRAGMENT(32768u); break;
    case 16: HAP_LAUNCH_FRAGMENT(65536u); break;
    default: return 1;
    }
#undef HAP_LAUNCH_FRAGMENT
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
