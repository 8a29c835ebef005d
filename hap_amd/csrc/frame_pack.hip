// frame_pack.hip -- device-side Hap frame assembly for gfx950.
//
// The reference writes a frame while it compresses, chunk after chunk, because chunk i's
// position depends on the compressed sizes of all chunks before it (hap.c:448-476,
// `compressed_data += chunk_packed_length`).  On the GPU every fragment is compressed at once
// into a worst-case slot, then:
//
//   frame_chunk_sums_kernel  one wavefront per chunk: sums its fragments' compressed sizes.
//   frame_pack_kernel    one workgroup per frame: decides per chunk "store raw iff compressed >=
//                        chunk size" (hap.c:460-471) and per texture "store the whole texture raw
//                        iff no gain" (hap.c:478-495), prefix-sums the chunk positions, and writes
//                        every header and table of the frame (hap.c:436-440, 497-501, 598) plus,
//                        optionally, the private fragment-size section 0x46.
//   frame_moves_kernel   one wavefront per chunk: one move per fragment (prefix sum of the sizes)
//                        and the per-fragment entries of the private table.
//   frame_gather_kernel  one wavefront per move: slot (or raw texture bytes) -> final position.
//
// No host round trip: the only thing the host reads back is bytes_used / status per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hapgpu_abi.h"

namespace {

__device__ __forceinline__ void put24(uint8_t *p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); }
__device__ __forceinline__ void put32(uint8_t *p, unsigned v) { put24(p, v); p[3] = (uint8_t)(v >> 24); }

// section header, reference hap.c:189-212
__device__ void write_section(uint8_t *p, unsigned header_len, unsigned length, unsigned type)
{
    if (header_len == 4u) {
        put24(p, length);
    } else {
        put24(p, 0u);
        put32(p + 4, length);
    }
    p[3] = (uint8_t)type;
}

__device__ __forceinline__ unsigned varint_len(unsigned v) { return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u; }

__device__ void write_varint(uint8_t *p, unsigned v)
{
    while (v >= 0x80u) {
        *p++ = (uint8_t)(v | 0x80u);
        v >>= 7;
    }
    *p = (uint8_t)v;
}

// block-wide exclusive scan of one 64-bit value per thread (256 threads); returns the exclusive
// prefix, *total receives the block sum.
__device__ unsigned long long block_scan(unsigned long long v, unsigned long long *total, unsigned long long *lds)
{
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d);
        if ((int)lane >= d)
            incl += up;
    }
    __syncthreads();
    if (lane == 63)
        lds[wave] = incl;
    __syncthreads();
    unsigned long long before = 0, sum = 0;
    for (unsigned w = 0; w < 4; w++) {
        if (w < wave)
            before += lds[w];
        sum += lds[w];
    }
    *total = sum;
    return before + incl - v;
}

// [device-only] what the three pack kernels hand to each other about one chunk
struct ChunkPack {
    uint64_t dst;        // where the chunk's stored bytes begin in the frame (kernel 2 -> 3)
    uint64_t itab;       // where its fragment-size entries go, 0 = no table (kernel 2 -> 3)
    uint32_t csize;      // varint + sum of its fragments' compressed sizes (kernel 1 -> 2)
    uint32_t how;        // 0: compressed chunk, 1: chunk stored raw, 2: whole texture stored raw (kernel 2 -> 3)
};

// 1. one wavefront per chunk: compressed size of the chunk = varint + sum over its fragments
__global__ __launch_bounds__(64) void frame_chunk_sums_kernel(const HapGpuFrameEnc *frames, const uint32_t *__restrict__ frag_sizes,
                                                              ChunkPack *__restrict__ packs, unsigned chunks_per_frame)
{
    const HapGpuFrameEnc &frame = frames[blockIdx.z];
    const unsigned t = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
    if (t >= frame.tex_count)
        return;
    const HapGpuTexEnc &tex = frame.tex[t];
    if (i >= tex.chunk_count)
        return;
    ChunkPack *pk = packs + (size_t)blockIdx.z * chunks_per_frame + (t ? frame.tex[0].chunk_count : 0u) + i;
    unsigned long long sum = 0;
    if (tex.compressor == 1u) {
        const uint32_t *fs = frag_sizes + tex.frag_first + (size_t)i * tex.frags_per_chunk;
        for (unsigned k = lane; k < tex.frags_per_chunk; k += 64u)
            sum += fs[k] & ~HAPGPU_FRAG_PUBLISHED;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
            sum += __shfl_xor(sum, d);
        sum += varint_len(tex.chunk_bytes);
    }
    if (lane == 0) {
        pk->csize = (uint32_t)(sum > 0xFFFFFFFFull ? 0xFFFFFFFFull : sum);
        pk->dst = 0;
        pk->itab = 0;
        pk->how = 2u;
    }
}

// 2. one workgroup per frame: store-raw decisions, chunk positions (prefix sums over the chunk sizes of kernel 1),
//    every header and table except the per-fragment entries
__global__ __launch_bounds__(256) void frame_pack_kernel(HapGpuFrameEnc *frames, unsigned frag_log2,
                                                         const uint8_t *__restrict__ group_tables,
                                                         HapGpuCopyEntry *__restrict__ copies, unsigned extra_first,
                                                         ChunkPack *__restrict__ packs, unsigned chunks_per_frame)
{
    __shared__ unsigned long long scan_lds[4];
    HapGpuFrameEnc &frame = frames[blockIdx.x];
    const unsigned tid = threadIdx.x;
    uint8_t *cursor = (uint8_t *)frame.dst + frame.outer_header_len;
    unsigned long long sections_total = 0;
    unsigned extra_at = extra_first + blockIdx.x * chunks_per_frame;      // this frame's group table moves
    ChunkPack *pk = packs + (size_t)blockIdx.x * chunks_per_frame;
    // placed streams (HapGpuTexEnc.reserved bit 27) lie where this kernel's sums put them only if every chunk shrank
    int not_placed = (frame.reserved & 1u) != 0u && ((frame.tex[0].reserved >> 27) & 1u) != 0u;

    for (unsigned t = 0; t < frame.tex_count; t++) {
        const HapGpuTexEnc tex = frame.tex[t];
        uint8_t *sec = cursor;
        const unsigned n = tex.chunk_count, fpc = tex.frags_per_chunk, cb = tex.chunk_bytes, hdr = tex.header_len;
        unsigned long long body = tex.bytes;
        bool complex_frame = false;

        if (tex.compressor == 1u) {
            const unsigned vlen = varint_len(cb);
            // fragment table version 3 (field streams): + a 96-byte group table per fragment
            const bool with_tiles = tex.emit_index && ((tex.reserved >> 20) & 1u) != 0u && group_tables != nullptr;
            const unsigned index_len = tex.emit_index ? 8u + (with_tiles ? 4u + HAP_GROUP_TABLE_BYTES : 4u) * n * fpc : 0u;
            const unsigned ilen = 5u * n + 8u + index_len;
            // pass 1: total stored payload
            unsigned long long total = 0;
            for (unsigned base = 0; base < n; base += 256u) {
                const unsigned i = base + tid;
                unsigned long long stored = 0;
                if (i < n) {
                    const unsigned long long c = pk[i].csize;
                    stored = c >= cb ? cb : c;                                   // hap.c:460-466
                }
                unsigned long long tile_total;
                block_scan(stored, &tile_total, scan_lds);
                total += tile_total;
            }
            const unsigned long long complex_body = 4ull + ilen + total;
            complex_frame = complex_body < (unsigned long long)tex.bytes + hdr;   // hap.c:478
            if (complex_frame) {
                body = complex_body;
                uint8_t *ctab = sec + hdr + 8u;
                uint8_t *stab = ctab + n + 4u;
                uint8_t *itab = stab + 4u * n;                                    // fragment section, if any
                uint8_t *payload = sec + hdr + 4u + ilen;
                if (tid == 0) {
                    write_section(sec + hdr, 4u, ilen, HAP_SECTION_INSTRUCTIONS);      // hap.c:436
                    write_section(sec + hdr + 4u, 4u, n, HAP_SECTION_COMPRESSORS);     // hap.c:438
                    write_section(ctab + n, 4u, 4u * n, HAP_SECTION_SIZES);            // hap.c:440
                    if (tex.emit_index) {
                        write_section(itab, 4u, index_len - 4u, HAP_SECTION_FRAGMENTS);
                        itab[4] = (uint8_t)(with_tiles ? HAP_FRAGMENT_TABLE_VERSION_FIELDS : HAP_FRAGMENT_TABLE_VERSION);
                        itab[5] = (uint8_t)frag_log2;
                        // granularity_log2 of the element streams (version 3: | block layout << 4;
                        // compressor code 4 -> 4 = [2,6,4,4], 10 -> 2 = [4,4], 2 -> 6 = [2,6], 12 -> 8 = [4,4,4,4])
                        const unsigned code = (tex.reserved >> 16) & 0xFu;
                        itab[6] = (uint8_t)((tex.reserved & 0xFu) | (with_tiles ? (code == 4u ? 4u : code == 10u ? 2u : code == 12u ? 8u : 6u) << 4 : 0u));
                        itab[7] = (uint8_t)(tex.reserved >> 8);   // match window in 256-byte units, 0 = whole fragment
                    }
                }
                // pass 2: positions and the per-chunk table entries
                unsigned long long run = 0;
                for (unsigned base = 0; base < n; base += 256u) {
                    const unsigned i = base + tid;
                    unsigned long long csize = 0, stored = 0;
                    if (i < n) {
                        csize = pk[i].csize;
                        stored = csize >= cb ? cb : csize;
                    }
                    unsigned long long tile_total;
                    const unsigned long long off = run + block_scan(stored, &tile_total, scan_lds);
                    run += tile_total;
                    if (i < n) {
                        const bool raw = csize >= cb;
                        if (raw && t == 0u && ((tex.reserved >> 27) & 1u))
                            not_placed = 1;
                        ctab[i] = raw ? (uint8_t)HAP_NIBBLE_NONE : (uint8_t)HAP_NIBBLE_SNAPPY;   // hap.c:465,470
                        put32(stab + 4u * i, (unsigned)stored);                                  // hap.c:472
                        uint8_t *at = payload + off;
                        if (!raw) {
                            write_varint(at, cb);
                            at += vlen;
                        }
                        pk[i].dst = (uint64_t)at;
                        pk[i].itab = tex.emit_index ? (uint64_t)(itab + 8u + 4u * (size_t)i * fpc) : 0u;
                        pk[i].how = raw ? 1u : 0u;
                        // the group tables follow the fragment sizes: one move per chunk (consecutive fragments)
                        HapGpuCopyEntry e;
                        e.reserved = 0;
                        e.src = (uint64_t)(group_tables + (size_t)(tex.frag_first + i * fpc) * HAP_GROUP_TABLE_BYTES);
                        e.dst = (uint64_t)(itab + 8u + 4u * n * fpc + (size_t)i * fpc * HAP_GROUP_TABLE_BYTES);
                        // (a placed texture's wavefronts have written their group tables here themselves)
                        e.len = with_tiles && !(t == 0u && ((tex.reserved >> 27) & 1u)) ? fpc * HAP_GROUP_TABLE_BYTES : 0u;
                        copies[extra_at + i] = e;
                    }
                }
            }
        }
        if (!complex_frame && t == 0u && tex.compressor == 1u && ((tex.reserved >> 27) & 1u))
            not_placed = 1;
        if (!complex_frame) {
            // whole texture stored as-is, reference hap.c:490-495
            for (unsigned i = tid; i < n; i += 256u) {
                HapGpuCopyEntry e;
                e.reserved = 0; e.src = 0; e.dst = 0; e.len = 0;
                copies[extra_at + i] = e;
                pk[i].dst = (uint64_t)(sec + hdr + (size_t)i * cb);
                pk[i].itab = 0;
                pk[i].how = 2u;
            }
        }
        if (tid == 0) {
            const unsigned type = ((complex_frame ? HAP_NIBBLE_COMPLEX : HAP_NIBBLE_NONE) << 4) | (tex.format_nibble & 0xFu);
            write_section(sec, hdr, (unsigned)body, type);                         // hap.c:499
        }
        cursor += hdr + body;
        sections_total += hdr + body;
        extra_at += n;
        pk += n;
        __syncthreads();
    }
    not_placed = __syncthreads_or(not_placed);
    if (tid == 0 && not_placed) {
        frame.bytes_used = 0;
        frame.status = HAPGPU_STATUS_NOT_PLACED;
    } else if (tid == 0) {
        if (frame.outer_header_len)
            write_section((uint8_t *)frame.dst, frame.outer_header_len, (unsigned)sections_total, HAP_SECTION_MULTI);   // hap.c:598
        frame.bytes_used = frame.outer_header_len + sections_total;
        frame.status = 0;
    }
}

// 3. one wavefront per chunk: one move per fragment (slot -> its place in the chunk: prefix sum over the fragment
//    sizes) and the fragment-size entries of the private table
__global__ __launch_bounds__(64) void frame_moves_kernel(const HapGpuFrameEnc *frames, unsigned frag_log2,
                                                         const uint8_t *__restrict__ slots, unsigned slot_stride,
                                                         const uint32_t *__restrict__ frag_sizes,
                                                         HapGpuCopyEntry *__restrict__ copies,
                                                         const ChunkPack *__restrict__ packs, unsigned chunks_per_frame)
{
    const HapGpuFrameEnc &frame = frames[blockIdx.z];
    const unsigned t = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
    if (t >= frame.tex_count)
        return;
    const HapGpuTexEnc &tex = frame.tex[t];
    if (i >= tex.chunk_count)
        return;
    const ChunkPack pk = packs[(size_t)blockIdx.z * chunks_per_frame + (t ? frame.tex[0].chunk_count : 0u) + i];
    const unsigned fpc = tex.frags_per_chunk, cb = tex.chunk_bytes, frag_bytes = 1u << frag_log2;
    const uint8_t *tsrc = (const uint8_t *)tex.src;
    const uint32_t *fs = frag_sizes + tex.frag_first + (size_t)i * fpc;
    const bool last_chunk = i + 1u == tex.chunk_count;
    unsigned long long run = 0;
    for (unsigned base = 0; base < fpc; base += 64u) {
        const unsigned k = base + lane;
        const unsigned begin = k << frag_log2;
        unsigned len = 0;
        if (k < fpc) {
            if (pk.how == 0u) {
                len = fs[k] & ~HAPGPU_FRAG_PUBLISHED;
            } else {
                len = begin < cb ? min(frag_bytes, cb - begin) : 0u;
                if (pk.how == 2u && last_chunk && k + 1u == fpc)      // bytes not divisible by the chunk count: keep the tail
                    len = tex.bytes - (i * cb + begin);
            }
        }
        unsigned long long incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(incl, d);
            if ((int)lane >= d)
                incl += up;
        }
        const unsigned long long at = run + incl - len;
        run += __shfl(incl, 63);
        if (k < fpc) {
            const unsigned f = tex.frag_first + i * fpc + k;
            HapGpuCopyEntry e;
            e.reserved = 0;
            e.src = pk.how == 0u ? (uint64_t)(slots + (size_t)f * slot_stride) : (uint64_t)(tsrc + (size_t)i * cb + begin);
            e.dst = pk.dst + at;
            // (a placed stream is where it belongs already)
            e.len = (t == 0u && ((tex.reserved >> 27) & 1u) && pk.how == 0u) ? 0u : len;
            copies[f] = e;
            if (pk.itab)
                put32((uint8_t *)pk.itab + 4u * k, pk.how == 0u ? len : 0u);
        }
    }
}

__global__ __launch_bounds__(64) void frame_gather_kernel(const HapGpuCopyEntry *__restrict__ copies, unsigned count)
{
    const unsigned lane = threadIdx.x;
    if (blockIdx.x >= count)
        return;
    const HapGpuCopyEntry e = copies[blockIdx.x];
    uint8_t *dst = (uint8_t *)e.dst;
    const uint8_t *src = (const uint8_t *)e.src;
    unsigned len = e.len;
    if (len == 0)
        return;
    // head: bring dst to 16-byte alignment
    const unsigned mis = (unsigned)((uintptr_t)dst & 15u);
    if (mis) {
        const unsigned head = min(16u - mis, len);
        if (lane < head)
            dst[lane] = src[lane];
        dst += head;
        src += head;
        len -= head;
    }
    const unsigned wide = len >> 4;
    const unsigned smis = (unsigned)((uintptr_t)src & 3u);
    if (smis == 0) {
        for (unsigned i = lane; i < wide; i += 64u) {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(src + ((size_t)i << 4));
            *reinterpret_cast<uint4 *>(dst + ((size_t)i << 4)) = make_uint4(s[0], s[1], s[2], s[3]);
        }
    } else {
        // source not dword aligned: read the 5 covering dwords and shift
        const uint32_t *base = reinterpret_cast<const uint32_t *>(src - smis);
        for (unsigned i = lane; i < wide; i += 64u) {
            const uint32_t *s = base + ((size_t)i << 2);
            const unsigned a = s[0], b = s[1], c = s[2], d = s[3], f = s[4];
            *reinterpret_cast<uint4 *>(dst + ((size_t)i << 4)) =
                make_uint4(__builtin_amdgcn_alignbyte(b, a, smis), __builtin_amdgcn_alignbyte(c, b, smis),
                           __builtin_amdgcn_alignbyte(d, c, smis), __builtin_amdgcn_alignbyte(f, d, smis));
        }
    }
    const unsigned done = wide << 4;
    if (done + lane < len)
        dst[done + lane] = src[done + lane];
}

} // namespace

// pack_scratch: frame_count * chunks_per_frame * hapgpu_pack_scratch_bytes_per_chunk() bytes of device memory
extern "C" unsigned hapgpu_pack_scratch_bytes_per_chunk(void) { return (unsigned)sizeof(ChunkPack); }

extern "C" int hapgpu_launch_frame_pack(HapGpuFrameEnc *frames, unsigned frame_count, unsigned frag_log2,
                                        const void *slots, unsigned slot_stride, const uint32_t *frag_sizes,
                                        const uint8_t *group_tables, HapGpuCopyEntry *copies, unsigned extra_first,
                                        unsigned chunks_per_frame, unsigned max_chunks_per_texture, unsigned textures,
                                        void *pack_scratch, hipStream_t stream)
{
    if (frame_count == 0)
        return 0;
    if (!pack_scratch || max_chunks_per_texture == 0 || textures == 0)
        return 1;
    ChunkPack *packs = (ChunkPack *)pack_scratch;
    const dim3 per_chunk(max_chunks_per_texture, textures, frame_count);
    hipLaunchKernelGGL(frame_chunk_sums_kernel, per_chunk, dim3(64), 0, stream, frames, frag_sizes, packs, chunks_per_frame);
    hipLaunchKernelGGL(frame_pack_kernel, dim3(frame_count), dim3(256), 0, stream, frames, frag_log2, group_tables, copies,
                       extra_first, packs, chunks_per_frame);
    hipLaunchKernelGGL(frame_moves_kernel, per_chunk, dim3(64), 0, stream, frames, frag_log2, (const uint8_t *)slots,
                       slot_stride, frag_sizes, copies, packs, chunks_per_frame);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int hapgpu_launch_frame_gather(const HapGpuCopyEntry *copies, unsigned count, hipStream_t stream)
{
    if (count == 0)
        return 0;
    hipLaunchKernelGGL(frame_gather_kernel, dim3(count), dim3(64), 0, stream, copies, count);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
