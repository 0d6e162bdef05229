// valu_rate_probe.hip — what one SIMD of gfx950 sustains per instruction kind, by waves per SIMD (round 4: is k_tile's vector work,
// 2 042 instructions per wave on integer / bit operations, priced at 2 or at 4 clocks per wave-instruction?).
// Every workgroup is 256 threads = one wave per SIMD of its CU; grid = CUs x occ workgroups, so a SIMD hosts `occ` waves.  Each wave
// runs ITER iterations of 32 instructions of one kind on 8 independent registers and reports its own s_memtime span; the host prints
// clocks per wave-instruction per SIMD = span / (ITER * 32 * occ) (median over waves) and the same from the launch's wall time.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_rate_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R32(OP) R8(OP) R8(OP) R8(OP) R8(OP)

template <int KIND>
__global__ __launch_bounds__(256) void k_probe(uint32_t *sink, unsigned long long *span, int iters, uint32_t seed) {
    uint32_t a[8], b = seed + threadIdx.x, c = seed ^ 0x55u;
    uint32_t s0 = seed, s1 = seed + 1u; // scalar operands
    double d1 = 1.0000001 + 1e-9 * threadIdx.x; unsigned long long u1 = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1u); uint32_t ldsaddr = (threadIdx.x & 63u) * 4u;
    uint32_t big0 = 0x9E3779B9u * (threadIdx.x + seed), big1 = 0x85EBCA6Bu * (threadIdx.x + 3u * seed);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (uint32_t)(i + 3) + threadIdx.x;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 s[20:21], %0, %1" : : "v"(b), "v"(c) : "vcc", "s20", "s21");
    if (KIND >= 78 && KIND <= 90) { for (int i = 0; i < 8; i += 2) *reinterpret_cast<double *>(&a[i]) = 1.5 + 0.001 * (threadIdx.x + i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
#define OP(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 1) {
#define OP(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 2) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 3) {
#define OP(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 4) {
#define OP(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 5) {
#define OP(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 6) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
            R32(OP)
#undef OP
        } else if (KIND == 7) {
#define OP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 8) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 9) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 10) {
#define OP(i) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 11) {
#define OP(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 12) {
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 13) {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 14) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 15) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])) : "v"(((unsigned long long)b << 32) | c));
            R32(OP)
#undef OP
        } else if (KIND == 16) { // scalar only
#define OP(i) asm volatile("s_and_b32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
            R32(OP)
#undef OP
        } else if (KIND == 17) { // one vector and one scalar instruction alternating (32 + 32)
#define OP(i) asm volatile("v_add_u32 %0, %2, %0\n\ts_and_b32 %1, %1, %3" : "+v"(a[i]), "+s"(s0) : "v"(b), "s"(s1) : "scc");
            R32(OP)
#undef OP
        } else if (KIND == 18) { // ds_bpermute (goes through the LDS pipeline)
#define OP(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(b));
            R8(OP)
#undef OP
        } else if (KIND == 19) {
#define OP(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 20) { // 64-bit shift
#define OP(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])));
            R32(OP)
#undef OP
        } else if (KIND == 21) { // v_readlane to a scalar (vector issue + scalar write)
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 22) { // min3 / max3
#define OP(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 23) { // SDWA byte select
#define OP(i) asm volatile("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 24) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 25) {
#define OP(i) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 26) {
#define OP(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 27) {
#define OP(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 28) {
#define OP(i) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 29) {
#define OP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 30) {
#define OP(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 31) {
#define OP(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 32) {
#define OP(i) asm volatile("v_cmp_eq_u32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
            R32(OP)
#undef OP
        } else if (KIND == 33) {
#define OP(i) asm volatile("v_cmp_lt_i16 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 34) {
#define OP(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 35) {
#define OP(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 36) {
#define OP(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 37) {
#define OP(i) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 38) {
#define OP(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 39) {
#define OP(i) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 40) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 41) {
#define OP(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 42) {
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 43) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 44) {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 45) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 46) {
#define OP(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
            R32(OP)
#undef OP
        } else if (KIND == 47) {
#define OP(i) asm volatile("v_and_b32 %0, 0x12345, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 48) {
#define OP(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 49) {
#define OP(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 50) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 51) {
#define OP(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 52) {
#define OP(i) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 53) {
#define OP(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 54) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 55) {
#define OP(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 56) {
#define OP(i) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 57) {
#define OP(i) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 58) {
#define OP(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 59) {
#define OP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 60) {
#define OP(i) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
            R32(OP)
#undef OP
        } else if (KIND == 61) {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 62) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 63) {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 64) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(big0), "v"(big1));
            R32(OP)
#undef OP
        } else if (KIND == 65) {
#define OP(i) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(big0), "v"(big1));
            R32(OP)
#undef OP
        } else if (KIND == 66) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(a[i]) : "v"(big0));
            R32(OP)
#undef OP
        } else if (KIND == 67) {
#define OP(i) asm volatile("v_lshrrev_b32 %0, 7, %1" : "=v"(a[i]) : "v"(big0));
            R32(OP)
#undef OP
        } else if (KIND == 68) {
#define OP(i) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(a[i]) : "v"(b), "v"(big0));
            R32(OP)
#undef OP
        } else if (KIND == 69) {
#define OP(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
            R32(OP)
#undef OP
        } else if (KIND == 70) {
#define OP(i) asm volatile("v_and_b32 %0, 15, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 71) {
#define OP(i) asm volatile("v_add_u32 %0, 3, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 72) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(s1));
            R32(OP)
#undef OP
        } else if (KIND == 73) {
#define OP(i) asm volatile("v_mov_b32 %0, 0x12345" : "=v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 74) {
#define OP(i) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 75) {
#define OP(i) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 76) {
#define OP(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 77) {
#define OP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a[i]) : "s"(s1), "v"(c));
            R32(OP)
#undef OP
        } else if (KIND == 78) {
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(d1));
            R32(OP)
#undef OP
        } else if (KIND == 79) {
#define OP(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(d1));
            R32(OP)
#undef OP
        } else if (KIND == 80) {
#define OP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(d1));
            R32(OP)
#undef OP
        } else if (KIND == 81) {
#define OP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])));
            R32(OP)
#undef OP
        } else if (KIND == 82) {
#define OP(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])));
            R32(OP)
#undef OP
        } else if (KIND == 83) {
#define OP(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])));
            R32(OP)
#undef OP
        } else if (KIND == 84) {
#define OP(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(d1) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 85) {
#define OP(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(d1));
            R32(OP)
#undef OP
        } else if (KIND == 86) {
#define OP(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(d1));
            R32(OP)
#undef OP
        } else if (KIND == 87) {
#define OP(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(*reinterpret_cast<double *>(&a[(i) & 6])), "v"(d1) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 88) {
#define OP(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 89) {
#define OP(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d1));
            R32(OP)
#undef OP
        } else if (KIND == 90) {
#define OP(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&a[(i) & 6])) : "v"(d1));
            R32(OP)
#undef OP
        } else if (KIND == 91) {
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])) : "v"(b), "v"(c) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 92) {
#define OP(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])) : "v"(b), "v"(c) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 93) {
#define OP(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        } else if (KIND == 94) {
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])) : "v"(u1));
            R32(OP)
#undef OP
        } else if (KIND == 95) {
#define OP(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])));
            R32(OP)
#undef OP
        } else if (KIND == 96) {
#define OP(i) asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])));
            R32(OP)
#undef OP
        } else if (KIND == 97) {
#define OP(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])), "v"(u1) : "vcc");
            R32(OP)
#undef OP
        } else if (KIND == 98) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<unsigned long long *>(&a[(i) & 6])) : "v"(u1));
            R32(OP)
#undef OP
        } else if (KIND == 99) {
#define OP(i) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a[i]) : "v"(ldsaddr));
            R32(OP)
#undef OP
        } else if (KIND == 100) {
#define OP(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(ldsaddr));
            R32(OP)
            asm volatile("s_waitcnt lgkmcnt(0)");
#undef OP
        } else if (KIND == 101) {
#define OP(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 102) {
#define OP(i) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(s1));
            R32(OP)
#undef OP
        } else if (KIND == 103) {
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 104) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 105) {
#define OP(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 106) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(big0));
            R32(OP)
#undef OP
        } else if (KIND == 107) {
#define OP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 108) {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
            R32(OP)
#undef OP
        } else if (KIND == 109) {
#define OP(i) asm volatile("v_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            R32(OP)
#undef OP
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t x = s0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i];
    if (x == 0x1234567u) sink[0] = x;
    if ((threadIdx.x & 63) == 0) span[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(uint32_t *, unsigned long long *, int, uint32_t);
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t *sink; unsigned long long *span;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&span, sizeof(unsigned long long) * 4 * cus * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *nm[110] = {"v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_bfe_u32", "v_ffbl_b32", "v_bcnt_u32_b32", "v_cndmask_b32", "v_cmp_lt_u32",
                          "v_mad_u32_u24", "v_mul_lo_u32", "v_pk_sub_u16", "v_add3_u32", "v_and_or_b32", "v_mov_b32_dpp", "v_fma_f32", "v_pk_fma_f32",
                          "s_and_b32", "v_add+s_and pair", "ds_bpermute+wait", "v_lshl_or_b32", "v_lshlrev_b64", "v_readlane_b32", "v_min3_u32", "v_min_u32_sdwa", "v_mov_b32", "v_or_b32", "v_xor_b32", "v_sub_u32", "v_not_b32", "v_bitop3_b32", "v_or3_b32", "v_lshl_add_u32", "v_cmp_eq_u32 e64 sgpr", "v_cmp_lt_i16", "v_lshrrev_b32", "v_ffbh_u32", "v_min_u32", "v_max_u32", "v_xad_u32", "v_bfi_b32", "v_perm_b32", "v_alignbit_b32", "v_mul_u32_u24", "v_cndmask vcc(set)", "v_cndmask e64 sgpr", "v_cndmask new dst", "v_add_u32 sgpr src", "v_and_b32 literal", "v_add_co_u32", "v_addc_co_u32", "v_mul_f32", "v_max_f32", "v_subrev_u32", "v_ashrrev_i32", "v_lshlrev_b32 vreg", "v_pk_add_u16", "v_add_u16", "v_sad_u32", "v_mbcnt_lo", "cmp vcc + cndmask e32", "cmp sgpr + cndmask e64", "cndmask e64 vcc", "cndmask e32 + v_add", "cndmask 0,1,vcc", "cndmask e32 large vals", "v_add_u32 large vals", "v_lshlrev_b32 imm 7", "v_lshrrev_b32 imm 7", "v_lshrrev_b32 vreg", "v_and_b32 sgpr src", "v_and_b32 inline const", "v_add_u32 inline const", "v_mov_b32 from sgpr", "v_mov_b32 literal", "v_cmp e32 vcc vgprs", "v_bfe_i32", "v_mul_i32_i24", "v_bitop3 sgpr src", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_cmp_lt_f64", "v_cvt_f64_i32", "v_cvt_i32_f64", "v_max_f64", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_hi_u32", "v_lshl_add_u64", "v_lshrrev_b64", "v_ashrrev_i64", "v_cmp_lt_u64", "v_pk_mul_f32", "ds_read_b32 (idx)", "ds_bpermute x8 then wait", "v_readfirstlane", "v_writelane", "v_cvt_f32_u32", "v_rcp_f32", "v_sqrt_f32", "v_mul_lo_u32 again", "v_add_u32_dpp row_shr", "v_mov_dpp row_bcast15", "v_swap / v_permlane? n/a -> v_max_u16"};
    kern_t ks[110] = {k_probe<0>, k_probe<1>, k_probe<2>, k_probe<3>, k_probe<4>, k_probe<5>, k_probe<6>, k_probe<7>, k_probe<8>, k_probe<9>, k_probe<10>, k_probe<11>,
                     k_probe<12>, k_probe<13>, k_probe<14>, k_probe<15>, k_probe<16>, k_probe<17>, k_probe<18>, k_probe<19>, k_probe<20>, k_probe<21>, k_probe<22>, k_probe<23>, k_probe<24>, k_probe<25>, k_probe<26>, k_probe<27>, k_probe<28>, k_probe<29>, k_probe<30>, k_probe<31>, k_probe<32>, k_probe<33>, k_probe<34>, k_probe<35>, k_probe<36>, k_probe<37>, k_probe<38>, k_probe<39>, k_probe<40>, k_probe<41>, k_probe<42>, k_probe<43>, k_probe<44>, k_probe<45>, k_probe<46>, k_probe<47>, k_probe<48>, k_probe<49>, k_probe<50>, k_probe<51>, k_probe<52>, k_probe<53>, k_probe<54>, k_probe<55>, k_probe<56>, k_probe<57>, k_probe<58>, k_probe<59>, k_probe<60>, k_probe<61>, k_probe<62>, k_probe<63>, k_probe<64>, k_probe<65>, k_probe<66>, k_probe<67>, k_probe<68>, k_probe<69>, k_probe<70>, k_probe<71>, k_probe<72>, k_probe<73>, k_probe<74>, k_probe<75>, k_probe<76>, k_probe<77>, k_probe<78>, k_probe<79>, k_probe<80>, k_probe<81>, k_probe<82>, k_probe<83>, k_probe<84>, k_probe<85>, k_probe<86>, k_probe<87>, k_probe<88>, k_probe<89>, k_probe<90>, k_probe<91>, k_probe<92>, k_probe<93>, k_probe<94>, k_probe<95>, k_probe<96>, k_probe<97>, k_probe<98>, k_probe<99>, k_probe<100>, k_probe<101>, k_probe<102>, k_probe<103>, k_probe<104>, k_probe<105>, k_probe<106>, k_probe<107>, k_probe<108>, k_probe<109>};
    const int occs[4] = {1, 2, 4, 8};
    printf("%-18s", "kind \\ waves/SIMD");
    for (int o = 0; o < 4; o++) printf("   occ=%d: s_memtime  wall@clk", occs[o]);
    printf("   (clocks per wave-instruction per SIMD)\n");
    for (int k = (argc > 2 ? atoi(argv[2]) : 0); k < 110; k++) {
        printf("%-40s", nm[k]);
        const int per_iter = k == 18 ? 8 : (k == 17 ? 32 : 32); // (kind 17: 32 pairs)
        for (int o = 0; o < 4; o++) {
            const int occ = occs[o], grid = cus * occ;
            float best = 1e9f;
            std::vector<unsigned long long> h(4 * grid);
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(ks[k], dim3(grid), dim3(256), 0, 0, sink, span, iters, (uint32_t)(rep + 7));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 1 && ms < best) best = ms;
            }
            CK(hipMemcpy(h.data(), span, sizeof(unsigned long long) * 4 * grid, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2];
            const double per = med / ((double)iters * per_iter * occ);
            const double wall = (double)best * 1e-3 * (double)prop.clockRate * 1e3 / ((double)iters * per_iter * occ);
            printf("   %9.2f %9.2f        ", per, wall);
        }
        printf("\n");
    }
    return 0;
}
