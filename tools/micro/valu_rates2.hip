// Dev probe 2: one opcode per kernel via inline asm (8 independent chains, 8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define KERNEL(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)                      \
    {                                                                                              \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, \
                 a6 = a0 ^ 99, a7 = a0 + 13;                                                       \
        unsigned b = seed | 1u;                                                                    \
        for (int i = 0; i < ITER; i++) {                                                           \
            asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)            \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21"); \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
    }
#define A_ADD(x) "v_add_u32 " #x ", " #x ", %8\n"
#define A_AND(x) "v_and_b32 " #x ", " #x ", %8\n"
#define A_LSHL(x) "v_lshlrev_b32 " #x ", 1, " #x "\n"
#define A_BFE(x) "v_bfe_u32 " #x ", " #x ", 3, 8\n"
#define A_LSHLOR(x) "v_lshl_or_b32 " #x ", " #x ", 2, %8\n"
#define A_MINU(x) "v_min_u32 " #x ", " #x ", %8\n"
#define A_MINI(x) "v_min_i32 " #x ", " #x ", %8\n"
#define A_MINF(x) "v_min_f32 " #x ", " #x ", %8\n"
#define A_ADDF(x) "v_add_f32 " #x ", " #x ", %8\n"
#define A_FMA(x) "v_fma_f32 " #x ", " #x ", %8, " #x "\n"
#define A_CVTUB(x) "v_cvt_f32_ubyte0 " #x ", " #x "\n"
#define A_CVTU(x) "v_cvt_u32_f32 " #x ", " #x "\n"
#define A_MAD24(x) "v_mad_u32_u24 " #x ", " #x ", %8, " #x "\n"
#define A_MADI24(x) "v_mad_i32_i24 " #x ", " #x ", %8, " #x "\n"
#define A_CND(x) "v_cndmask_b32 " #x ", " #x ", %8, vcc\n"
#define A_CND64(x) "v_cndmask_b32_e64 " #x ", " #x ", %8, s[20:21]\n"
#define A_CMPCND(x) "v_cmp_lt_u32 vcc, " #x ", %8\nv_cndmask_b32 " #x ", " #x ", %8, vcc\n"
#define A_CMPCND64(x) "v_cmp_lt_u32_e64 s[20:21], " #x ", %8\nv_cndmask_b32_e64 " #x ", " #x ", %8, s[20:21]\n"
#define A_CMP(x) "v_cmp_lt_u32 vcc, " #x ", %8\n"
#define A_CMPF(x) "v_cmp_lt_f32 vcc, " #x ", %8\n"
#define A_MOV(x) "v_mov_b32 " #x ", %8\n"
#define A_XOR(x) "v_xor_b32 " #x ", " #x ", %8\n"
#define A_MIN3(x) "v_min3_u32 " #x ", " #x ", %8, " #x "\n"
#define A_MED3F(x) "v_med3_f32 " #x ", " #x ", %8, " #x "\n"
#define A_SAD(x) "v_sad_u32 " #x ", " #x ", %8, " #x "\n"
#define A_PKMIN(x) "v_pk_min_u16 " #x ", " #x ", %8\n"
#define A_PKFMA(x) "v_pk_fma_f32 " #x ", " #x ", " #x ", " #x "\n"
#define A_DOT4(x) "v_dot4_u32_u8 " #x ", " #x ", %8, " #x "\n"
#define A_FFBL(x) "v_ffbl_b32 " #x ", " #x "\n"
#define A_MBCNT(x) "v_mbcnt_lo_u32_b32 " #x ", " #x ", %8\n"
#define A_ADD3(x) "v_add3_u32 " #x ", " #x ", %8, " #x "\n"
KERNEL(k_add, A_ADD) KERNEL(k_and, A_AND) KERNEL(k_lshl, A_LSHL) KERNEL(k_bfe, A_BFE) KERNEL(k_lshlor, A_LSHLOR)
KERNEL(k_minu, A_MINU) KERNEL(k_mini, A_MINI) KERNEL(k_minf, A_MINF) KERNEL(k_addf, A_ADDF) KERNEL(k_fma, A_FMA)
KERNEL(k_cvtub, A_CVTUB) KERNEL(k_cvtu, A_CVTU) KERNEL(k_mad24, A_MAD24) KERNEL(k_madi24, A_MADI24) KERNEL(k_cnd, A_CND)
KERNEL(k_cmp, A_CMP) KERNEL(k_cmpf, A_CMPF) KERNEL(k_mov, A_MOV) KERNEL(k_xor, A_XOR) KERNEL(k_min3, A_MIN3)
KERNEL(k_med3f, A_MED3F) KERNEL(k_sad, A_SAD) KERNEL(k_pkmin, A_PKMIN) KERNEL(k_dot4, A_DOT4) KERNEL(k_ffbl, A_FFBL)
KERNEL(k_mbcnt, A_MBCNT) KERNEL(k_add3, A_ADD3) KERNEL(k_cnd64, A_CND64) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_cmpcnd64, A_CMPCND64)
template <typename K> void run(const char *name, K kern, unsigned *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = blocks * 4.0 / 1024.0 * ITER * 8;
    printf("%-22s %7.3f ms  %.2f ns/instr/SIMD (%.2f cyc @2.4GHz)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
#define RUN(n) run(#n, n, d);
    RUN(k_add) RUN(k_and) RUN(k_xor) RUN(k_lshl) RUN(k_bfe) RUN(k_lshlor) RUN(k_add3) RUN(k_minu) RUN(k_mini) RUN(k_min3) RUN(k_sad) RUN(k_mad24) RUN(k_madi24)
    RUN(k_dot4) RUN(k_pkmin) RUN(k_ffbl) RUN(k_mbcnt) RUN(k_mov) RUN(k_cnd) RUN(k_cnd64) RUN(k_cmpcnd) RUN(k_cmpcnd64) RUN(k_cmp) RUN(k_cmpf) RUN(k_minf) RUN(k_addf) RUN(k_fma) RUN(k_med3f) RUN(k_cvtub) RUN(k_cvtu)
    return 0;
}
