// tools/micro/pinned_copy.hip — where does the host write path's upload time go?  CPU memcpy pageable -> pinned (by flags, by threads),
// H2D of the pinned piece, D2H + CPU memcpy pinned -> pageable.  hipcc -O2 -o pinned_copy pinned_copy.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t CH = 32u << 20, N = 64;                 // 64 pieces of 32 MiB = 2 GiB per direction
    const size_t ENT = 65536;
    std::vector<char*> srcs;                              // separate pageable buffers, as a caller's files would be
    for (size_t i = 0; i < CH * 8 / ENT; i++) { char* p = (char*)malloc(ENT); memset(p, (int)i, ENT); srcs.push_back(p); }
    char* dev; hipMalloc((void**)&dev, CH * 2);
    hipStream_t st; hipStreamCreate(&st);
    struct { const char* name; unsigned flags; } kinds[] = { {"default", hipHostMallocDefault}, {"non-coherent", hipHostMallocNonCoherent}, {"write-combined", hipHostMallocWriteCombined},
                                                              {"numa-user|non-coherent", hipHostMallocNumaUser | hipHostMallocNonCoherent} };
    for (auto& k : kinds) {
        char* pin = nullptr;
        if (hipHostMalloc((void**)&pin, CH, k.flags) != hipSuccess) { printf("%-24s: hipHostMalloc refused\n", k.name); (void)hipGetLastError(); continue; }
        memset(pin, 1, CH);
        for (int T : {1, 4, 8}) {
            auto part = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) memcpy(pin + i * ENT, srcs[i % srcs.size()], ENT); };
            double t0 = now();
            for (size_t r = 0; r < N; r++) {
                std::vector<std::thread> th;
                const size_t cnt = CH / ENT;
                for (int t = 1; t < T; t++) th.emplace_back(part, cnt * t / T, cnt * (t + 1) / T);
                part(0, cnt / T);
                for (auto& x : th) x.join();
            }
            double t1 = now();
            printf("%-24s gather pageable->pinned, %d thread(s): %.1f GB/s\n", k.name, T, N * CH / (t1 - t0) / 1e9);
        }
        double t0 = now();
        for (size_t r = 0; r < N; r++) hipMemcpyAsync(dev, pin, CH, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        double t1 = now();
        printf("%-24s H2D from it: %.1f GB/s\n", k.name, N * CH / (t1 - t0) / 1e9);
        t0 = now();
        for (size_t r = 0; r < N; r++) hipMemcpyAsync(pin, dev, CH, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        t1 = now();
        printf("%-24s D2H into it: %.1f GB/s\n", k.name, N * CH / (t1 - t0) / 1e9);
        {
            std::vector<char*> outs; for (size_t i = 0; i < CH / ENT; i++) outs.push_back((char*)malloc(ENT));
            for (auto p : outs) memset(p, 0, ENT);
            for (int T : {1, 4}) {
                auto part = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) memcpy(outs[i], pin + i * ENT, ENT); };
                double a = now();
                for (size_t r = 0; r < N; r++) {
                    std::vector<std::thread> th; const size_t cnt = CH / ENT;
                    for (int t = 1; t < T; t++) th.emplace_back(part, cnt * t / T, cnt * (t + 1) / T);
                    part(0, cnt / T);
                    for (auto& x : th) x.join();
                }
                double b = now();
                printf("%-24s scatter pinned->pageable, %d thread(s): %.1f GB/s\n", k.name, T, N * CH / (b - a) / 1e9);
            }
            for (auto p : outs) free(p);
        }
        hipHostFree(pin);
    }
    // pageable straight to the device, one copy per 32 MiB (the driver stages)
    char* big = (char*)malloc(CH); memset(big, 3, CH);
    double t0 = now();
    for (size_t r = 0; r < N; r++) hipMemcpyAsync(dev, big, CH, hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);
    printf("pageable 32 MiB pieces straight H2D: %.1f GB/s\n", N * CH / (now() - t0) / 1e9);
    return 0;
}
