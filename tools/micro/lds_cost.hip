// lds_cost.hip — what does one wave64 LDS instruction of each kind cost the CU's LDS pipe on gfx950, with the address
// patterns the LZ4 / Zstandard decoders actually use (byte reads at random positions, 16-byte reads and writes at any byte
// offset, ds_bpermute), at the kernels' occupancy (8 waves per SIMD, 64-thread workgroups)?  Developer microbenchmark;
// not part of the product.  Prints shader cycles per wave-instruction per CU (s_memtime span / instructions per CU).
//   hipcc --offload-arch=gfx950 -O3 -o lds_cost lds_cost.hip && ./lds_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#define LDS __attribute__((address_space(3)))
#define REP8(x) x x x x x x x x

enum Mode { RD_U8_LINEAR, RD_U8_RANDOM, RD_U8_WALK60, RD_U16_ODD, RD_B32_RANDOM, RD2_B32_RANDOM, RD_B64_ALIGNED, RD_B64_ANY, RD_B128_ALIGNED,
            RD_B128_ANY, BPERMUTE, SWIZZLE, WR_B8_RANDOM, WR_B16_ANY, WR_B32_ANY, WR_B64_ANY, WR_B128_ALIGNED, WR_B128_ANY, OR_B64_ALIGNED,
            VALU_DPP, RD_U8_LINEAR_X3, MODES };
static const char* names[MODES] = {
    "ds_read_u8   lane*4 (no conflict)", "ds_read_u8   random byte", "ds_read_u8   60*lane + rnd(60) (walk)", "ds_read_u16  random odd address",
    "ds_read_b32  random dword", "ds_read2_b32 random dword pair", "ds_read_b64  8*lane+rnd*8 aligned", "ds_read_b64  random byte offset",
    "ds_read_b128 16*lane aligned", "ds_read_b128 random byte offset", "ds_bpermute_b32 random lane", "ds_swizzle_b32", "ds_write_b8  random byte",
    "ds_write_b16 random byte offset", "ds_write_b32 random byte offset", "ds_write_b64 random byte offset", "ds_write_b128 16*lane aligned",
    "ds_write_b128 ~13*lane+rnd (any offset)", "ds_or_b64    random aligned", "v_add_u32 dpp row_shr:1 (VALU reference)", "ds_read_u8 sorted ~7 B apart (token fetch)" };

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(64, 8) void k(uint32_t* out, unsigned long long* span, int iters, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint8_t sh[4096 + 64];
    const int lane = threadIdx.x;
    for (int j = lane; j < 4096 + 64; j += 64) sh[j] = (uint8_t)(j * 7 + seed);
    __syncthreads();
    LDS uint8_t* base = (LDS uint8_t*)sh;
    uint32_t s = seed * 977u + lane * 131u + blockIdx.x, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        uint32_t r0 = rnd(s);
        uint32_t a;
        switch (MODE) {
        case RD_U8_LINEAR: a = lane * 4 + (r0 & 3); break;
        case RD_U8_RANDOM: a = r0 & 4095; break;
        case RD_U8_WALK60: a = 60 * lane + (r0 % 60); break;
        case RD_U16_ODD: a = (r0 & 4094) | 1; break;
        case RD_B32_RANDOM: case RD2_B32_RANDOM: a = (r0 & 4092); break;
        case RD_B64_ALIGNED: case OR_B64_ALIGNED: a = (r0 & 4088); break;
        case RD_B64_ANY: case WR_B64_ANY: case WR_B32_ANY: case WR_B16_ANY: a = r0 & 4095; break;
        case RD_B128_ALIGNED: case WR_B128_ALIGNED: a = 16 * lane + (r0 & 0xC00); break;
        case RD_B128_ANY: a = r0 & 4095; break;
        case WR_B128_ANY: a = 13 * lane + (r0 & 7) + (r0 & 0x800); break;
        case WR_B8_RANDOM: a = r0 & 4095; break;
        case RD_U8_LINEAR_X3: a = 7 * lane + (r0 & 3) + (r0 & 0x600); break;
        default: a = (r0 & 63) * 4; break;
        }
        LDS uint8_t* p = base + a;
        uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0;
        uint32_t w[4] = {r0, r0 + 1, r0 + 2, r0 + 3};
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u4 q0, q1; u2 d0, d1; (void)q0; (void)q1; (void)d0; (void)d1;
        // eight independent instructions, then one wait
        if (MODE == RD_U8_LINEAR || MODE == RD_U8_RANDOM || MODE == RD_U8_WALK60 || MODE == RD_U8_LINEAR_X3) {
            asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:1\n ds_read_u8 %2, %8 offset:2\n ds_read_u8 %3, %8 offset:3\n"
                         "ds_read_u8 %4, %8 offset:4\n ds_read_u8 %5, %8 offset:5\n ds_read_u8 %6, %8 offset:6\n ds_read_u8 %7, %8 offset:7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(p) : "memory");
        } else if (MODE == RD_U16_ODD) {
            asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:2\n ds_read_u16 %2, %8 offset:4\n ds_read_u16 %3, %8 offset:6\n"
                         "ds_read_u16 %4, %8 offset:8\n ds_read_u16 %5, %8 offset:10\n ds_read_u16 %6, %8 offset:12\n ds_read_u16 %7, %8 offset:14\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(p) : "memory");
        } else if (MODE == RD_B32_RANDOM) {
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                         "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(p) : "memory");
        } else if (MODE == RD2_B32_RANDOM) {
            u2 e0, e1, e2, e3;
            asm volatile("ds_read2_b32 %0, %8 offset1:1\n ds_read2_b32 %1, %8 offset0:2 offset1:3\n ds_read2_b32 %2, %8 offset0:4 offset1:5\n ds_read2_b32 %3, %8 offset0:6 offset1:7\n"
                         "ds_read2_b32 %4, %8 offset0:8 offset1:9\n ds_read2_b32 %5, %8 offset0:10 offset1:11\n ds_read2_b32 %6, %8 offset0:12 offset1:13\n ds_read2_b32 %7, %8 offset0:14 offset1:15\n s_waitcnt lgkmcnt(0)"
                         : "=v"(d0), "=v"(d1), "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3), "=v"(*(u2*)&w[0]), "=v"(*(u2*)&w[2]) : "v"(p) : "memory");
            v0 = d0.x ^ d1.y ^ e0.x ^ e1.y ^ e2.x ^ e3.y ^ w[0] ^ w[3];
        } else if (MODE == RD_B64_ALIGNED || MODE == RD_B64_ANY) {
            u2 e0, e1, e2, e3, e4, e5;
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                         "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)"
                         : "=v"(d0), "=v"(d1), "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3), "=v"(e4), "=v"(e5) : "v"(p) : "memory");
            v0 = d0.x ^ d1.y ^ e0.x ^ e1.y ^ e2.x ^ e3.y ^ e4.x ^ e5.y;
        } else if (MODE == RD_B128_ALIGNED || MODE == RD_B128_ANY) {
            u4 f0, f1, f2, f3, f4, f5;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8\n ds_read_b128 %5, %8 offset:16\n ds_read_b128 %6, %8 offset:32\n ds_read_b128 %7, %8 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=v"(q0), "=v"(q1), "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3), "=v"(f4), "=v"(f5) : "v"(p) : "memory");
            v0 = q0.x ^ q1.y ^ f0.z ^ f1.w ^ f2.x ^ f3.y ^ f4.z ^ f5.w;
        } else if (MODE == BPERMUTE) {
            const uint32_t idx = (r0 & 63) * 4;
            asm volatile("ds_bpermute_b32 %0, %8, %9\n ds_bpermute_b32 %1, %8, %10\n ds_bpermute_b32 %2, %8, %11\n ds_bpermute_b32 %3, %8, %12\n"
                         "ds_bpermute_b32 %4, %8, %9\n ds_bpermute_b32 %5, %8, %10\n ds_bpermute_b32 %6, %8, %11\n ds_bpermute_b32 %7, %8, %12\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(idx), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "memory");
        } else if (MODE == SWIZZLE) {
            asm volatile("ds_swizzle_b32 %0, %8 offset:0x041F\n ds_swizzle_b32 %1, %9 offset:0x081F\n ds_swizzle_b32 %2, %10 offset:0x101F\n ds_swizzle_b32 %3, %11 offset:0x041F\n"
                         "ds_swizzle_b32 %4, %8 offset:0x081F\n ds_swizzle_b32 %5, %9 offset:0x101F\n ds_swizzle_b32 %6, %10 offset:0x041F\n ds_swizzle_b32 %7, %11 offset:0x081F\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "memory");
        } else if (MODE == WR_B8_RANDOM) {
            asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %2 offset:1\n ds_write_b8 %0, %3 offset:2\n ds_write_b8 %0, %4 offset:3\n"
                         "ds_write_b8 %0, %1 offset:4\n ds_write_b8 %0, %2 offset:5\n ds_write_b8 %0, %3 offset:6\n ds_write_b8 %0, %4 offset:7\n s_waitcnt lgkmcnt(0)"
                         : : "v"(p), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "memory");
        } else if (MODE == WR_B16_ANY) {
            asm volatile("ds_write_b16 %0, %1\n ds_write_b16 %0, %2 offset:3\n ds_write_b16 %0, %3 offset:6\n ds_write_b16 %0, %4 offset:9\n"
                         "ds_write_b16 %0, %1 offset:12\n ds_write_b16 %0, %2 offset:15\n ds_write_b16 %0, %3 offset:18\n ds_write_b16 %0, %4 offset:21\n s_waitcnt lgkmcnt(0)"
                         : : "v"(p), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "memory");
        } else if (MODE == WR_B32_ANY) {
            asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:5\n ds_write_b32 %0, %3 offset:10\n ds_write_b32 %0, %4 offset:15\n"
                         "ds_write_b32 %0, %1 offset:20\n ds_write_b32 %0, %2 offset:25\n ds_write_b32 %0, %3 offset:30\n ds_write_b32 %0, %4 offset:35\n s_waitcnt lgkmcnt(0)"
                         : : "v"(p), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "memory");
        } else if (MODE == WR_B64_ANY) {
            u2 x; x.x = w[0]; x.y = w[1];
            asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:9\n ds_write_b64 %0, %1 offset:18\n ds_write_b64 %0, %1 offset:27\n"
                         "ds_write_b64 %0, %1 offset:36\n ds_write_b64 %0, %1 offset:45\n ds_write_b64 %0, %1 offset:54\n ds_write_b64 %0, %1 offset:63\n s_waitcnt lgkmcnt(0)"
                         : : "v"(p), "v"(x) : "memory");
        } else if (MODE == WR_B128_ALIGNED || MODE == WR_B128_ANY) {
            u4 x; x.x = w[0]; x.y = w[1]; x.z = w[2]; x.w = w[3];
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n"
                         "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)"
                         : : "v"(p), "v"(x) : "memory");
        } else if (MODE == OR_B64_ALIGNED) {
            u2 x; x.x = w[0]; x.y = w[1];
            asm volatile("ds_or_b64 %0, %1\n ds_or_b64 %0, %1 offset:8\n ds_or_b64 %0, %1 offset:16\n ds_or_b64 %0, %1 offset:24\n"
                         "ds_or_b64 %0, %1 offset:32\n ds_or_b64 %0, %1 offset:40\n ds_or_b64 %0, %1 offset:48\n ds_or_b64 %0, %1 offset:56\n s_waitcnt lgkmcnt(0)"
                         : : "v"(p), "v"(x) : "memory");
        } else if (MODE == VALU_DPP) {
            asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
            v0 = w[0] ^ w[1] ^ w[2] ^ w[3];
        }
        acc += v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc + sh[acc & 4095];
    if (lane == 0) { atomicMin(&span[0], t0); atomicMax(&span[1], t1); }
}

// which lane's bytes survive when the 16-byte stores of ONE ds_write_b128 overlap (lane k writes at 6*k)?
__global__ void k_overlap(uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t sh[1024];
    const int lane = threadIdx.x;
    for (int j = lane; j < 1024; j += 64) sh[j] = 0xFF;
    __syncthreads();
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 x; x.x = x.y = x.z = x.w = 0x01010101u * (uint32_t)lane;
    LDS uint8_t* p = (LDS uint8_t*)sh + 6 * lane;
    asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(p), "v"(x) : "memory");
    __syncthreads();
    for (int j = lane; j < 512; j += 64) out[j] = sh[j];
}

template <int MODE> static void run(uint32_t* d, unsigned long long* span)
{
    const int cus = 256, wps = 8, blocks = cus * 4 * wps, iters = 2000;
    unsigned long long init[2] = {~0ull, 0ull};
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, span, 10, 1u);
    hipDeviceSynchronize();
    hipMemcpy(span, init, sizeof(init), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, span, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long sp[2]; hipMemcpy(sp, span, sizeof(sp), hipMemcpyDeviceToHost);
    const double cyc = (double)(sp[1] - sp[0]);                       // shader cycles, first wave start to last wave end
    const double per_cu = 8.0 * iters * 4 * wps;                      // wave-instructions per CU
    printf("%-46s %8.3f ms  %7.2f cycles per wave-instruction per CU   (%.2f GHz effective)\n", names[MODE], ms, cyc / per_cu, cyc / (ms * 1e6));
}

int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    unsigned long long* span; hipMalloc(&span, 16);
    run<RD_U8_LINEAR>(d, span); run<RD_U8_RANDOM>(d, span); run<RD_U8_WALK60>(d, span); run<RD_U8_LINEAR_X3>(d, span); run<RD_U16_ODD>(d, span);
    run<RD_B32_RANDOM>(d, span); run<RD2_B32_RANDOM>(d, span); run<RD_B64_ALIGNED>(d, span); run<RD_B64_ANY>(d, span);
    run<RD_B128_ALIGNED>(d, span); run<RD_B128_ANY>(d, span); run<BPERMUTE>(d, span); run<SWIZZLE>(d, span);
    run<WR_B8_RANDOM>(d, span); run<WR_B16_ANY>(d, span); run<WR_B32_ANY>(d, span); run<WR_B64_ANY>(d, span);
    run<WR_B128_ALIGNED>(d, span); run<WR_B128_ANY>(d, span); run<OR_B64_ALIGNED>(d, span); run<VALU_DPP>(d, span);
    // overlap order, three launches: is it stable?
    uint32_t h[512];
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_overlap, dim3(1), dim3(64), 0, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int hi_wins = 0, lo_wins = 0, other = 0;
        for (int b = 0; b < 6 * 63; b++) {                              // byte b may be written by lanes ceil((b-15)/6) .. b/6
            const int hi = b / 6, lo = (b - 15 + 5) / 6 < 0 ? 0 : (b - 15 + 5) / 6;
            if (hi == lo) continue;
            if ((int)h[b] == hi) hi_wins++; else if ((int)h[b] == lo) lo_wins++; else other++;
        }
        printf("overlapping ds_write_b128 (lane k at byte 6k): highest lane wins %d, lowest %d, another %d\n", hi_wins, lo_wins, other);
    }
    return 0;
}
