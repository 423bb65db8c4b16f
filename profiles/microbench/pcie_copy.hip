// pcie_copy.hip -- what the host <-> device link of the box gives a single process (hipcc --offload-arch=gfx950 -O2 -lpthread):
// pinned and pageable, both directions, 1.13 GB in 32 MiB pieces (the staging buffer size of crp_api.cpp), plus the
// host-side memcpy rates (1..16 threads) that feed / drain the pinned buffers.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void pcopy(char *dst, const char *src, size_t n, int t)
{
    std::vector<std::thread> pool;
    const size_t per = (n + t - 1) / t;
    for (int k = 0; k < t; ++k) {
        const size_t a = std::min(n, per * k), b = std::min(n, a + per);
        if (a < b) pool.emplace_back([=] { std::memcpy(dst + a, src + a, b - a); });
    }
    for (auto &th : pool) th.join();
}

int main()
{
    const size_t N = 1130ull << 20, CH = 32ull << 20;
    char *d = nullptr, *pin = nullptr;
    hipMalloc(&d, N);
    hipHostMalloc(&pin, N, hipHostMallocDefault);
    char *page = static_cast<char *>(malloc(N));
    memset(page, 1, N);
    memset(pin, 2, N);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (size_t o = 0; o < N; o += CH) hipMemcpyAsync(d + o, pin + o, std::min(CH, N - o), hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        printf("H2D pinned   %.1f GB/s\n", N / (now() - t0) / 1e9);
        t0 = now();
        for (size_t o = 0; o < N; o += CH) hipMemcpyAsync(pin + o, d + o, std::min(CH, N - o), hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        printf("D2H pinned   %.1f GB/s\n", N / (now() - t0) / 1e9);
        t0 = now();
        hipMemcpy(d, page, N, hipMemcpyHostToDevice);
        printf("H2D pageable %.1f GB/s\n", N / (now() - t0) / 1e9);
        t0 = now();
        hipMemcpy(page, d, N, hipMemcpyDeviceToHost);
        printf("D2H pageable %.1f GB/s (touched destination)\n", N / (now() - t0) / 1e9);
        char *fresh = static_cast<char *>(malloc(N));
        t0 = now();
        hipMemcpy(fresh, d, N, hipMemcpyDeviceToHost);
        printf("D2H pageable %.1f GB/s (fresh destination)\n", N / (now() - t0) / 1e9);
        free(fresh);
    }
    for (int t : {1, 2, 4, 8, 16}) {
        double t0 = now();
        pcopy(pin, page, N, t);
        const double a = N / (now() - t0) / 1e9;
        char *fresh = static_cast<char *>(malloc(N));
        t0 = now();
        pcopy(fresh, pin, N, t);
        printf("memcpy %2d threads: pageable->pinned %.1f GB/s, pinned->fresh pageable %.1f GB/s\n", t, a, N / (now() - t0) / 1e9);
        free(fresh);
    }
    // first touch of a fresh destination: page faults one by one (memcpy above) against MADV_POPULATE_WRITE per slice
    for (int t : {1, 4, 8, 16}) {
        char *fresh = static_cast<char *>(malloc(N));
        double t0 = now();
        std::vector<std::thread> pool;
        const size_t per = ((N + t - 1) / t + 4095) & ~(size_t)4095;
        const uintptr_t base = (reinterpret_cast<uintptr_t>(fresh) + 4095) & ~(uintptr_t)4095;
        const size_t usable = N - (base - reinterpret_cast<uintptr_t>(fresh)) - 4096;
        int rc_all = 0;
        for (int k = 0; k < t; ++k) {
            const size_t a = std::min(usable, per * k), b = std::min(usable, a + per);
            if (a < b) pool.emplace_back([=, &rc_all] { if (madvise(reinterpret_cast<void *>(base + a), b - a, 23 /* MADV_POPULATE_WRITE */)) rc_all = 1; });
        }
        for (auto &th : pool) th.join();
        const double pop = N / (now() - t0) / 1e9;
        t0 = now();
        hipMemcpy(fresh, d, N, hipMemcpyDeviceToHost);
        printf("populate %2d threads: %.1f GB/s (rc %d), then D2H into it %.1f GB/s\n", t, pop, rc_all, N / (now() - t0) / 1e9);
        free(fresh);
    }
    FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[128] = {0};
    if (f && fgets(line, sizeof line, f)) printf("THP: %s", line);
    return 0;
}
