// Dev probe: issue cost of the integer VALU instructions the hot kernels are made of.
// Each kernel runs ITER x 8 independent chains per lane of one opcode, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, a6 = a0 ^ 99, a7 = a0 + 13;
    const unsigned b = seed | 1u;
    unsigned long long w0 = a0, w1 = a1;
#pragma unroll 1
    for (int i = 0; i < ITER; i++) {
#define R8(F) a0 = F(a0); a1 = F(a1); a2 = F(a2); a3 = F(a3); a4 = F(a4); a5 = F(a5); a6 = F(a6); a7 = F(a7);
        if (OP == 0) {
#define F0(x) ((x) + b)
            R8(F0)
        } else if (OP == 1) {
#define F1(x) __builtin_amdgcn_udot4(x, b, x, false)
            R8(F1)
        } else if (OP == 2) {
#define F2(x) ((x) > b ? (x) - b : (x) + 3u)
            R8(F2)
        } else if (OP == 3) {
#define F3(x) __builtin_amdgcn_perm(x, b, x)
            R8(F3)
        } else if (OP == 4) {
#define F4(x) ((x) * b)
            R8(F4)
        } else if (OP == 5) {
#define F5(x) __umul24(x, b)
            R8(F5)
        } else if (OP == 6) {
            w0 = (w0 >> (a0 & 31)) + b; w1 = (w1 >> (a1 & 31)) + b; a0 += (unsigned)w0; a1 += (unsigned)w1;
            w0 = (w0 >> (a0 & 31)) + b; w1 = (w1 >> (a1 & 31)) + b; a0 += (unsigned)w0; a1 += (unsigned)w1;
        } else if (OP == 7) {
#define F7(x) min(x, b) 
            R8(F7)
            a0 += i; a1 += i; a2 += i; a3 += i; a4 += i; a5 += i; a6 += i; a7 += i;
        } else if (OP == 8) {
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define F8(x) __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, x) + __builtin_bit_cast(us2, b)))
            R8(F8)
        } else if (OP == 9) {
#define F9(x) __builtin_amdgcn_alignbyte(x, b, x)
            R8(F9)
        } else if (OP == 10) {
            float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
            float f4 = __uint_as_float(a4), f5 = __uint_as_float(a5), f6 = __uint_as_float(a6), f7 = __uint_as_float(a7);
            const float fb = __uint_as_float(b);
            f0 = fmaf(f0, fb, f0); f1 = fmaf(f1, fb, f1); f2 = fmaf(f2, fb, f2); f3 = fmaf(f3, fb, f3);
            f4 = fmaf(f4, fb, f4); f5 = fmaf(f5, fb, f5); f6 = fmaf(f6, fb, f6); f7 = fmaf(f7, fb, f7);
            a0 = __float_as_uint(f0); a1 = __float_as_uint(f1); a2 = __float_as_uint(f2); a3 = __float_as_uint(f3);
            a4 = __float_as_uint(f4); a5 = __float_as_uint(f5); a6 = __float_as_uint(f6); a7 = __float_as_uint(f7);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)w0 + (unsigned)w1;
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
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d, 8); run<10>("v_fma_f32", d, 8); run<1>("v_dot4_u32_u8", d, 8); run<2>("cmp+sub+add+cndmask (4)", d, 32);
    run<3>("v_perm_b32", d, 8); run<4>("v_mul_lo_u32", d, 8); run<5>("v_mul_u32_u24", d, 8);
    run<6>("lshr_b64+add (x4, ~4 ops each)", d, 16); run<7>("v_min_u32 (+add)", d, 16); run<8>("v_pk_add_u16", d, 8); run<9>("v_alignbyte", d, 8);
    return 0;
}
