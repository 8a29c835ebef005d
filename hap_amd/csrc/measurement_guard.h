/*
 * measurement_guard.h -- the tuning switches of the kernel sources belong to measurement builds only.
 *
 * tools/build_variants.sh compiles A/B variants of the library with extra -D switches (LDS per wave, hop sets, workgroup
 * shapes, host-side phase traces) into hap_amd/variants/libhap_amd_<name>.so and defines HAP_MEASUREMENT_BUILD while it
 * does.  A stray -D of one of them in HIPFLAGS / CFLAGS of the product build stops the compilation here instead of
 * shipping a library that differs from the tested one without saying so.  None of the switches left in the
 * sources changes the bytes the library writes; the ablations of round 4 that did are no longer in the tree.
 */
#ifndef HAP_AMD_MEASUREMENT_GUARD_H
#define HAP_AMD_MEASUREMENT_GUARD_H

#if !defined(HAP_MEASUREMENT_BUILD) && ( \
    defined(SDF_BUF_BYTES) || defined(SDF_DYN_LDS) || defined(SDF_HOPS) || defined(BRK_TIMING) || defined(SDF_ONLY) || \
    defined(SDF_UNSAFE) || defined(SDF_ABL_NOROUNDS) || defined(SDF_ABL_NOPRODREADS) || defined(SDF_ABL_NORINGSTORE) || \
    defined(SCB_MIN_WAVES) || defined(SCB_PREFETCH) || defined(SCB_ONLY_FUSED_YCOCG) || defined(PLC_ABL) || defined(PLC_NO_INTERLEAVE) || \
    defined(HAP_BLK_FAR_FIRST) || defined(HAP_WG_WAVES) || defined(HAP_WG_SUBS) || defined(HAP_WG_HASH_BITS) || \
    defined(HAP_V2_IN_BYTES) || defined(HAP_V2_OWNER_BYTES) || defined(HAP_CHAIN_ROUNDS) || defined(HAPB_TRACE))
#error "a measurement switch is defined without HAP_MEASUREMENT_BUILD: build variants with tools/build_variants.sh, never the product library"
#endif

#endif
