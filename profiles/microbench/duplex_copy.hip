// duplex_copy.hip -- does the host link of this box carry H2D and D2H at the same time?
// Pinned host buffers, two streams: H2D alone, D2H alone, both together (chunks of 32 MiB queued back to back).
// Build: hipcc -O2 --offload-arch=gfx950 duplex_copy.hip -o duplex_copy ; prints one JSON line.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));        \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const size_t total = (argc > 1 ? std::atoll(argv[1]) : 1024) << 20, chunk = 32ull << 20;
    void *h_up, *h_down, *d_up, *d_down;
    CK(hipHostMalloc(&h_up, total, hipHostMallocDefault));
    CK(hipHostMalloc(&h_down, total, hipHostMallocDefault));
    CK(hipMalloc(&d_up, total));
    CK(hipMalloc(&d_down, total));
    std::memset(h_up, 1, total);
    CK(hipMemset(d_down, 2, total));
    hipStream_t s_up, s_down;
    CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
    auto run = [&](bool up, bool down) -> double {
        const double t0 = now();
        for (size_t off = 0; off < total; off += chunk) {
            if (up) (void)hipMemcpyAsync((char *)d_up + off, (char *)h_up + off, chunk, hipMemcpyHostToDevice, s_up);
            if (down) (void)hipMemcpyAsync((char *)h_down + off, (char *)d_down + off, chunk, hipMemcpyDeviceToHost, s_down);
        }
        (void)hipStreamSynchronize(s_up);
        (void)hipStreamSynchronize(s_down);
        return now() - t0;
    };
    run(true, true);  // warm
    double up = 1e9, down = 1e9, both = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        up = std::min(up, run(true, false));
        down = std::min(down, run(false, true));
        both = std::min(both, run(true, true));
    }
    // the same bytes on FOUR streams that each carry both directions in turn (a lane per slice: up, up, down, ...)
    hipStream_t lane[4];
    for (auto &l : lane) CK(hipStreamCreateWithFlags(&l, hipStreamNonBlocking));
    auto run_lanes = [&]() -> double {
        const double t0 = now();
        size_t k = 0;
        for (size_t off = 0; off < total; off += chunk, ++k) {
            hipStream_t s = lane[(k / 2) % 4];  // two chunks per "slice"
            (void)hipMemcpyAsync((char *)d_up + off, (char *)h_up + off, chunk, hipMemcpyHostToDevice, s);
            if (k % 2 == 1) (void)hipMemcpyAsync((char *)h_down + off - chunk, (char *)d_down + off - chunk, 2 * chunk, hipMemcpyDeviceToHost, s);
        }
        for (auto &l : lane) (void)hipStreamSynchronize(l);
        return now() - t0;
    };
    run_lanes();
    double lanes4 = 1e9;
    for (int rep = 0; rep < 5; ++rep) lanes4 = std::min(lanes4, run_lanes());
    std::printf("{\"mixed_direction_streams\": 4, \"both_s\": %.4f, \"aggregate_GBs\": %.1f}\n", lanes4, 2 * total / 1e9 / lanes4);
    const double gb = total / 1e9;
    std::printf("{\"bytes_each_way\": %zu, \"h2d_alone_GBs\": %.1f, \"d2h_alone_GBs\": %.1f, \"both_s\": %.4f, \"both_each_way_GBs\": %.1f, "
                "\"both_aggregate_GBs\": %.1f, \"full_duplex_would_be_s\": %.4f}\n",
                total, gb / up, gb / down, both, gb / both, 2 * gb / both, std::max(up, down));
    return 0;
}
