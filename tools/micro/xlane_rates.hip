// Dev probe: cost of the cross-lane moves the field-stream decoder can be built from (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/xlane_rates.hip -o tools/micro/xlane_rates && tools/micro/xlane_rates
// Each kernel runs ITER x 8 independent chains per lane of one operation, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    __shared__ unsigned lds[4096];
    const unsigned lane = threadIdx.x & 63u;
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, a6 = a0 ^ 99, a7 = a0 + 13;
    for (unsigned i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 4u + (seed & 3u) * 4u;   // values are valid byte addresses
    __syncthreads();
    // permutation addresses: byte address of a lane (x4)
    unsigned p0 = ((lane * 5u + 1u) & 63u) * 4u, p1 = ((lane + 17u) & 63u) * 4u, p2 = ((lane ^ 21u) & 63u) * 4u, p3 = ((lane * 3u) & 63u) * 4u;
    unsigned r0 = (threadIdx.x * 4u) & 16383u, r1 = (r0 + 1024u) & 16383u, r2 = (r0 + 2048u) & 16383u, r3 = (r0 + 3072u) & 16383u;
#pragma unroll 1
    for (int i = 0; i < ITER; i++) {
#define R8(F) a0 = F(a0); a1 = F(a1); a2 = F(a2); a3 = F(a3); a4 = F(a4); a5 = F(a5); a6 = F(a6); a7 = F(a7);
        if (OP == 0) {
            a0 = __builtin_amdgcn_ds_bpermute(p0, a0); a1 = __builtin_amdgcn_ds_bpermute(p1, a1);
            a2 = __builtin_amdgcn_ds_bpermute(p2, a2); a3 = __builtin_amdgcn_ds_bpermute(p3, a3);
            a4 = __builtin_amdgcn_ds_bpermute(p0, a4); a5 = __builtin_amdgcn_ds_bpermute(p1, a5);
            a6 = __builtin_amdgcn_ds_bpermute(p2, a6); a7 = __builtin_amdgcn_ds_bpermute(p3, a7);
        } else if (OP == 1) {          // self-addressed bpermute, as the decoder does (address = the value itself)
            a0 = __builtin_amdgcn_ds_bpermute(a0, a0); a1 = __builtin_amdgcn_ds_bpermute(a1, a1);
            a2 = __builtin_amdgcn_ds_bpermute(a2, a2); a3 = __builtin_amdgcn_ds_bpermute(a3, a3);
            a4 = __builtin_amdgcn_ds_bpermute(a4, a4); a5 = __builtin_amdgcn_ds_bpermute(a5, a5);
            a6 = __builtin_amdgcn_ds_bpermute(a6, a6); a7 = __builtin_amdgcn_ds_bpermute(a7, a7);
        } else if (OP == 2) {          // ds_read_b32, consecutive lanes consecutive dwords (values chain the addresses)
            a0 = lds[(r0 + (a0 & 4u)) >> 2]; a1 = lds[(r1 + (a1 & 4u)) >> 2]; a2 = lds[(r2 + (a2 & 4u)) >> 2]; a3 = lds[(r3 + (a3 & 4u)) >> 2];
            a4 = lds[(r0 + (a4 & 4u)) >> 2]; a5 = lds[(r1 + (a5 & 4u)) >> 2]; a6 = lds[(r2 + (a6 & 4u)) >> 2]; a7 = lds[(r3 + (a7 & 4u)) >> 2];
        } else if (OP == 3) {
#define F3(x) __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true) + 1u
            R8(F3)
        } else if (OP == 4) {
#define F4(x) __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, true) + 1u    /* wave_shr:1 */
            R8(F4)
        } else if (OP == 5) {
#define F5(x) __builtin_amdgcn_ds_swizzle(x, 0x8055) + 1u
            R8(F5)
        } else if (OP == 6) {          // dpp hop as the decoder does it: mov_dpp + cmp + cndmask
#define F6(x) ((x) == seed ? (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true) : (x))
            R8(F6)
            a0 += i; a1 += i; a2 += i; a3 += i; a4 += i; a5 += i; a6 += i; a7 += i;
        } else if (OP == 7) {
#define F7(x) __builtin_amdgcn_mov_dpp(x, 0x142, 0xF, 0xF, true) + 1u          /* row_bcast:15 */
            R8(F7)
        } else if (OP == 8) {
#define F8(x) ((x) + 1u)
            R8(F8)
        } else if (OP == 9) {          // v_permlane32_swap: lanes 32..63 of one register <-> lanes 0..31 of another
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
            a0 += 1u; a2 += 1u; a4 += 1u; a6 += 1u;
        } else if (OP == 10) {
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
            a0 += 1u; a2 += 1u; a4 += 1u; a6 += 1u;
        } else if (OP == 11) {         // ds_read2_b32 of two consecutive dwords at 16-byte lane stride (the ring read pattern)
            const uint2 v0 = *reinterpret_cast<const uint2 *>(&lds[((lane * 16u + (a0 & 8u)) >> 2) & 4095u]);
            const uint2 v1 = *reinterpret_cast<const uint2 *>(&lds[((lane * 16u + 1024u + (a1 & 8u)) >> 2) & 4095u]);
            const uint2 v2 = *reinterpret_cast<const uint2 *>(&lds[((lane * 16u + 2048u + (a2 & 8u)) >> 2) & 4095u]);
            const uint2 v3 = *reinterpret_cast<const uint2 *>(&lds[((lane * 16u + 3072u + (a3 & 8u)) >> 2) & 4095u]);
            a0 = v0.x ^ v0.y; a1 = v1.x ^ v1.y; a2 = v2.x ^ v2.y; a3 = v3.x ^ v3.y;
        } else if (OP == 12) {         // ds_or_b32 (no return) at scattered words
            atomicOr(&lds[(a0 >> 2) & 127u], 1u << (i & 31)); atomicOr(&lds[(a1 >> 2) & 127u], 1u << (i & 31));
            atomicOr(&lds[(a2 >> 2) & 127u], 1u << (i & 31)); atomicOr(&lds[(a3 >> 2) & 127u], 1u << (i & 31));
            a0 += 52u; a1 += 20u; a2 += 36u; a3 += 12u;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char *name, unsigned *d, int ops_per_iter)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;       // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double instr_per_simd = waves_per_simd * ITER * ops_per_iter;
    printf("%-44s %8.3f ms  -> %6.2f cycles per wave-instruction per SIMD, %6.2f per CU (2.4 GHz)\n", name, ms,
           ms * 1e6 / instr_per_simd * 2.4, ms * 1e6 / instr_per_simd * 2.4 / 4.0);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<8>("v_add_u32", d, 8);
    run<0>("ds_bpermute_b32 (fixed permutation)", d, 8);
    run<1>("ds_bpermute_b32 (self-addressed)", d, 8);
    run<2>("ds_read_b32 consecutive", d, 8);
    run<11>("ds_read2_b32 at 16-byte lane stride", d, 4);
    run<12>("ds_or_b32 scattered", d, 4);
    run<5>("ds_swizzle_b32", d, 8);
    run<3>("v_mov_dpp row_shr:1 (+add)", d, 16);
    run<4>("v_mov_dpp wave_shr:1 (+add)", d, 16);
    run<7>("v_mov_dpp row_bcast:15 (+add)", d, 16);
    run<6>("dpp hop: mov_dpp+cmp+cndmask (+add)", d, 32);
    run<9>("v_permlane32_swap (+add)", d, 8);
    run<10>("v_permlane16_swap (+add)", d, 8);
    return 0;
}
