// launch_sync.hip -- what one "launch a kernel, wait for it" costs the host, three ways of waiting:
//   hipStreamSynchronize with the default device flags, with hipDeviceScheduleSpin, and polling a pinned
//   host word the kernel writes last.  hipcc --offload-arch=gfx950 -O2 launch_sync.hip -o launch_sync && ./launch_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ void spin_kernel(uint64_t ticks, volatile uint64_t *done, uint64_t seq)
{
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        *done = seq;
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0 default flags, 1 spin flag, 2 poll pinned word
    const uint64_t ticks = argc > 2 ? strtoull(argv[2], 0, 10) : 0;
    if (mode == 1) hipSetDeviceFlags(hipDeviceScheduleSpin);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint64_t *done;
    hipHostMalloc(&done, 8, hipHostMallocDefault);
    *done = 0;
    uint64_t *ddone;
    hipHostGetDevicePointer((void **)&ddone, done, 0);
    const int N = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        for (int i = 1; i <= N; ++i) {
            const uint64_t seq = (uint64_t)rep * N + i;
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s, ticks, ddone, seq);
            if (mode == 2) {
                while (*(volatile uint64_t *)done != seq) {}
            } else {
                hipStreamSynchronize(s);
            }
        }
        const double dt = now() - t0;
        printf("mode %d ticks %llu: %.2f us per launch+wait\n", mode, (unsigned long long)ticks, dt / N * 1e6);
    }
    hipStreamSynchronize(s);
    return 0;
}
