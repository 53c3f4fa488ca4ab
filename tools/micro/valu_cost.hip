// valu_cost.hip — issue cost of the integer vector instructions the byte codecs are made of, on gfx950 at 8 waves per SIMD
// (64-thread workgroups): SIMD cycles per wave64 instruction at a nominal 2.4 GHz.  Developer microbenchmark; not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define OPS(X) \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0") \
    X(1, "v_lshlrev_b32", "v_lshlrev_b32 %0, %1, %0\n v_lshlrev_b32 %1, %2, %1\n v_lshlrev_b32 %2, %3, %2\n v_lshlrev_b32 %3, %0, %3") \
    X(2, "v_and_b32", "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0") \
    X(3, "v_bfe_u32", "v_bfe_u32 %0, %0, %1, 8\n v_bfe_u32 %1, %1, %2, 8\n v_bfe_u32 %2, %2, %3, 8\n v_bfe_u32 %3, %3, %0, 8") \
    X(4, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, %2\n v_alignbit_b32 %1, %1, %2, %3\n v_alignbit_b32 %2, %2, %3, %0\n v_alignbit_b32 %3, %3, %0, %1") \
    X(5, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, %2\n v_alignbyte_b32 %1, %1, %2, %3\n v_alignbyte_b32 %2, %2, %3, %0\n v_alignbyte_b32 %3, %3, %0, %1") \
    X(6, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1") \
    X(7, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0") \
    X(8, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0") \
    X(9, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1") \
    X(10, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %1, %1, 2, %2\n v_lshl_add_u32 %2, %2, 2, %3\n v_lshl_add_u32 %3, %3, 2, %0") \
    X(11, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1") \
    X(12, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1") \
    X(13, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %3, %3, %0, %1") \
    X(14, "v_lshlrev_b64", "v_lshlrev_b64 %4, %0, %4\n v_lshlrev_b64 %5, %1, %5\n v_lshlrev_b64 %4, %2, %4\n v_lshlrev_b64 %5, %3, %5") \
    X(15, "v_lshrrev_b64", "v_lshrrev_b64 %4, %0, %4\n v_lshrrev_b64 %5, %1, %5\n v_lshrrev_b64 %4, %2, %4\n v_lshrrev_b64 %5, %3, %5") \
    X(16, "v_cmp_lt_u32 + v_cndmask_b32", "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %0, vcc") \
    X(17, "v_cmp_lt_u32 sgpr pair + s_and_b64", "v_cmp_lt_u32 s[10:11], %0, %1\n s_and_b64 s[12:13], s[10:11], exec\n v_cmp_lt_u32 s[14:15], %1, %2\n s_and_b64 s[12:13], s[14:15], s[12:13]") \
    X(18, "v_readlane_b32", "v_readlane_b32 s10, %0, 3\n v_readlane_b32 s11, %1, 5\n v_readlane_b32 s12, %2, 7\n v_readlane_b32 s13, %3, 9") \
    X(19, "v_readfirstlane_b32", "v_readfirstlane_b32 s10, %0\n v_readfirstlane_b32 s11, %1\n v_readfirstlane_b32 s12, %2\n v_readfirstlane_b32 s13, %3") \
    X(20, "v_ffbl_b32", "v_ffbl_b32 %0, %1\n v_ffbl_b32 %1, %2\n v_ffbl_b32 %2, %3\n v_ffbl_b32 %3, %0") \
    X(21, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %3, %0, %3") \
    X(22, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_lo_u32_b32 %1, %2, %1\n v_mbcnt_lo_u32_b32 %2, %3, %2\n v_mbcnt_lo_u32_b32 %3, %0, %3") \
    X(23, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(24, "v_add_u32 dpp row_bcast:15", "v_add_u32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %3, %0, %3 row_bcast:15 row_mask:0xa bank_mask:0xf") \
    X(25, "v_lshlrev_b16 / sdwa byte select", "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_and_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_and_b32_sdwa %3, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") \
    X(26, "s_add_u32 (SALU reference)", "s_add_u32 s10, s10, s11\n s_xor_b32 s11, s11, s12\n s_add_u32 s12, s12, s13\n s_xor_b32 s13, s13, s10") \
    X(27, "v_add_co_u32 + v_addc_co_u32 (64-bit add)", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_add_co_u32 %1, vcc, %1, %0\n v_addc_co_u32 %3, vcc, %3, %2, vcc") \
    X(28, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0") \
    X(29, "v_max_u32 / v_min_u32", "v_max_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_min_u32 %3, %3, %0") \
    X(30, "v_max3_u32", "v_max3_u32 %0, %0, %1, %2\n v_max3_u32 %1, %1, %2, %3\n v_max3_u32 %2, %2, %3, %0\n v_max3_u32 %3, %3, %0, %1") \
    X(31, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0") \
    X(32, "v_or_b32", "v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %0") \
    X(33, "v_xor_b32", "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0") \
    X(34, "v_sub_u32", "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0") \
    X(35, "v_mov_b32", "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0") \
    X(36, "v_cndmask_b32 (vcc fixed)", "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc") \
    X(37, "v_cmp_lt_u32 vcc", "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0") \
    X(38, "v_lshrrev_b32", "v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %1, %2, %1\n v_lshrrev_b32 %2, %3, %2\n v_lshrrev_b32 %3, %0, %3") \
    X(39, "v_lshrrev_b32 by constant", "v_lshrrev_b32 %0, 4, %1\n v_lshrrev_b32 %1, 4, %2\n v_lshrrev_b32 %2, 4, %3\n v_lshrrev_b32 %3, 4, %0") \
    X(40, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 2, %1\n v_lshl_or_b32 %1, %1, 2, %2\n v_lshl_or_b32 %2, %2, 2, %3\n v_lshl_or_b32 %3, %3, 2, %0") \
    X(41, "v_add_co_u32 (carry out only)", "v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %1, vcc, %1, %2\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %3, vcc, %3, %0") \
    X(42, "v_addc_co_u32", "v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %0, vcc") \
    X(43, "v_lshl_add_u64", "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %4\n v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %4") \
    X(44, "v_mad_u64_u32", "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %5, vcc, %2, %3, %5\n v_mad_u64_u32 %4, vcc, %1, %2, %4\n v_mad_u64_u32 %5, vcc, %3, %0, %5") \
    X(45, "v_not_b32", "v_not_b32 %0, %1\n v_not_b32 %1, %2\n v_not_b32 %2, %3\n v_not_b32 %3, %0") \
    X(46, "v_and_b32 with literal constant", "v_and_b32 %0, 0xff00ff, %1\n v_and_b32 %1, 0xff00ff, %2\n v_and_b32 %2, 0xff00ff, %3\n v_and_b32 %3, 0xff00ff, %0") \
    X(47, "v_add_u32 e64 (sgpr operand)", "v_add_u32 %0, s10, %1\n v_add_u32 %1, s11, %2\n v_add_u32 %2, s12, %3\n v_add_u32 %3, s13, %0") \
    X(48, "v_mov_b64", "v_mov_b64 %4, %5\n v_mov_b64 %5, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %4") \
    X(49, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %1\n v_ashrrev_i32 %1, 3, %2\n v_ashrrev_i32 %2, 3, %3\n v_ashrrev_i32 %3, 3, %0") \
    X(50, "v_subrev_u32 / v_sub_co", "v_subrev_u32 %0, %0, %1\n v_subrev_u32 %1, %1, %2\n v_subrev_u32 %2, %2, %3\n v_subrev_u32 %3, %3, %0") \
    X(51, "v_pk_add_u16 / pk ops", "v_pk_lshlrev_b16 %0, 1, %1\n v_pk_lshlrev_b16 %1, 1, %2\n v_pk_lshlrev_b16 %2, 1, %3\n v_pk_lshlrev_b16 %3, 1, %0") \
    X(52, "v_cmp_eq_u32 to sgpr pair", "v_cmp_eq_u32 s[10:11], %0, %1\n v_cmp_eq_u32 s[12:13], %1, %2\n v_cmp_eq_u32 s[14:15], %2, %3\n v_cmp_eq_u32 s[10:11], %3, %0") \
    X(53, "v_cndmask_b32 e64 (sgpr pair mask)", "v_cndmask_b32 %0, %0, %1, s[10:11]\n v_cndmask_b32 %1, %1, %2, s[12:13]\n v_cndmask_b32 %2, %2, %3, s[14:15]\n v_cndmask_b32 %3, %3, %0, s[10:11]")

template <int MODE>
__global__ __launch_bounds__(64, 8) void k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 0x55, d = c + 7;
    uint64_t e = a * 77ull + 5, f = b * 91ull + 3;
    for (int i = 0; i < iters; i++) {
#define X(id, name, txt) if (MODE == id) { REP16(asm volatile(txt : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "vcc", "scc", "s10", "s11", "s12", "s13", "s14", "s15");) }
        OPS(X)
#undef X
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + (uint32_t)e + (uint32_t)f;
}

template <int MODE> static void run(const char* name, uint32_t* d)
{
    const int cus = 256, wps = 8, blocks = cus * 4 * wps, iters = 1500;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 64.0 * iters * wps;
    printf("%-44s %7.3f ms  %6.2f SIMD-cycles per wave64 instruction (8 waves/SIMD, nominal 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}

int main()
{
    uint32_t* d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
#define X(id, name, txt) run<id>(name, d);
    OPS(X)
#undef X
    return 0;
}
