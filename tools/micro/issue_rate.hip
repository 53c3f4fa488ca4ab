// issue_rate.hip — how many VALU / SALU / mixed instructions does one CU of gfx950 issue per cycle at the LZ4 kernel's
// occupancy (7 waves per SIMD, 64-thread workgroups)?  Developer microbenchmark; not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(64, 7) void k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 0x55, d = c + 7;
    uint32_t s0 = seed, s1 = seed * 5 + 1, s2 = seed ^ 3, s3 = seed + 11;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // 64 VALU
            REP16(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (MODE == 1) { // 64 SALU
            REP16(asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_xor_b32 %3, %3, %0" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else if (MODE == 2) { // 64 VALU + 64 SALU interleaved
            REP16(asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_xor_b32 %1, %1, %2\n s_xor_b32 %5, %5, %6\n v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, %7\n v_xor_b32 %3, %3, %0\n s_xor_b32 %7, %7, %4"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else if (MODE == 3) { // 64 VALU + 16 SALU
            REP16(asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_xor_b32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_xor_b32 %3, %3, %0"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else if (MODE == 4) { // divergent-if pattern: v_cmp + s_and_saveexec + v_add + s_or exec, x16
            REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[10:11], vcc\n v_add_u32 %2, %2, %3\n s_or_b64 exec, exec, s[10:11]\n v_xor_b32 %0, %0, %2\n v_add_u32 %1, %1, 1"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "s10", "s11", "scc");)
        } else if (MODE == 5) { // the same with v_cndmask instead of exec masking
            REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_add_u32 %3, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n v_xor_b32 %0, %0, %2\n v_add_u32 %1, %1, 1"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)
        } else if (MODE == 6) { // LDS reads u8 dependent chain + VALU (walk-like): 16 x (ds_read_u8, wait, 3 valu)
            __shared__ uint8_t sh[4096];
            if (i == 0) { for (int j = threadIdx.x; j < 4096; j += 64) sh[j] = (uint8_t)(j * 7 + seed); }
            REP16(a = (a + sh[a & 4095] + 3) ; b ^= a;)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + s0 + s1 + s2 + s3;
}
template <int MODE> static void run(const char* name, int per_iter, uint32_t* d, int waves_per_simd)
{
    const int cus = 256, blocks = cus * 4 * waves_per_simd, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)per_iter * iters * waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-34s waves/SIMD %d: %.3f ms  -> %.3f instr/cycle/SIMD (%.2f /cycle/CU) at a nominal 2.4 GHz\n", name, waves_per_simd, ms, instr_per_simd / cyc, 4 * instr_per_simd / cyc);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    for (int w : {1, 2, 4, 7}) {
        run<0>("64 VALU", 64, d, w);
        run<1>("64 SALU", 64, d, w);
        run<2>("64 VALU + 64 SALU", 128, d, w);
        run<3>("64 VALU + 16 SALU", 80, d, w);
        run<4>("exec-masked if x16 (6 instr)", 96, d, w);
        run<5>("cndmask if x16 (5 instr)", 80, d, w);
        run<6>("lds chain x16", 16, d, w);
    }
    return 0;
}
