// Are unaligned LDS reads (ds_read_b32/b64/b128 at arbitrary byte addresses) correct and fast on gfx950?
// hipcc -O3 --offload-arch=gfx950 tools/micro/lds_unaligned.hip -o /tmp/lds_unaligned && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
struct __attribute__((packed)) P128 { uint32_t a, b, c, d; };
struct __attribute__((packed)) P64 { uint32_t a, b; };
struct __attribute__((packed)) P32 { uint32_t a; };

__device__ __forceinline__ uint32_t asm_b32(unsigned addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ uint64_t asm_b64(unsigned addr)
{
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <int MODE>   // 0: packed b128, 1: 5 aligned dwords + alignbyte, 2: asm b64 x2, 3: asm b32 x4
__global__ void k(const uint8_t *in, uint32_t *out, unsigned shift, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[8192 + 64];
    for (unsigned i = threadIdx.x; i < 8192 + 64; i += blockDim.x) s[i] = in[i];
    __syncthreads();
    unsigned acc = 0;
    unsigned o = (threadIdx.x * 16u + shift) & 8191u;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            P128 v = *reinterpret_cast<const P128 *>(s + o);
            acc += v.a ^ v.b ^ v.c ^ v.d;
        } else if (MODE == 1) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(s) + (o >> 2);
            const unsigned sh = o & 3u;
            const unsigned a0 = w[0], a1 = w[1], a2 = w[2], a3 = w[3], a4 = w[4];
            acc += __builtin_amdgcn_alignbyte(a1, a0, sh) ^ __builtin_amdgcn_alignbyte(a2, a1, sh) ^
                   __builtin_amdgcn_alignbyte(a3, a2, sh) ^ __builtin_amdgcn_alignbyte(a4, a3, sh);
        } else if (MODE == 2) {
            const unsigned base = (unsigned)(uintptr_t)s;   // LDS address = low 32 bits of the generic pointer
            const uint64_t v = asm_b64(base + o), u = asm_b64(base + o + 8);
            acc += (uint32_t)v ^ (uint32_t)(v >> 32) ^ (uint32_t)u ^ (uint32_t)(u >> 32);
        } else {
            const unsigned base = (unsigned)(uintptr_t)s;
            acc += asm_b32(base + o) ^ asm_b32(base + o + 4) ^ asm_b32(base + o + 8) ^ asm_b32(base + o + 12);
        }
        o = (o + 1040u + (acc & 0u)) & 8191u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run(const uint8_t *din, uint32_t *dout, const std::vector<uint8_t> &h, const char *name)
{
    const int iters = 2000, blocks = 2048, threads = 256;
    for (unsigned shift = 0; shift < 16; shift += (shift < 4 ? 1 : 3)) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k<MODE><<<blocks, threads>>>(din, dout, shift, iters);
        hipEventRecord(a);
        k<MODE><<<blocks, threads>>>(din, dout, shift, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<uint32_t> got(threads);
        hipMemcpy(got.data(), dout, threads * 4, hipMemcpyDeviceToHost);
        // host check
        int bad = 0;
        for (int t = 0; t < threads; t++) {
            unsigned acc = 0, o = (t * 16u + shift) & 8191u;
            for (int it = 0; it < iters; it++) {
                uint32_t w[4];
                memcpy(w, &h[o], 16);
                acc += w[0] ^ w[1] ^ w[2] ^ w[3];
                o = (o + 1040u) & 8191u;
            }
            bad += acc != got[t];
        }
        printf("%-28s shift %2u: %.3f ms  (%.1f G 16-byte reads/s)  %s\n", name, shift, ms,
               (double)blocks * threads * iters / ms / 1e6, bad ? "WRONG" : "ok");
    }
}

int main()
{
    std::vector<uint8_t> h(8192 + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 131u + (i >> 5) * 7u + 3u);
    uint8_t *din; uint32_t *dout;
    hipMalloc(&din, h.size()); hipMalloc(&dout, 2048 * 256 * 4);
    hipMemcpy(din, h.data(), h.size(), hipMemcpyHostToDevice);
    run<0>(din, dout, h, "packed ds_read_b128");
    run<1>(din, dout, h, "5 x b32 + 4 x alignbyte");
    run<2>(din, dout, h, "asm 2 x ds_read_b64");
    run<3>(din, dout, h, "asm 4 x ds_read_b32");
    return 0;
}
