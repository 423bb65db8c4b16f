// fetch_calibration.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on MI355X for access patterns OTHER than the
// wide coalesced stream they are calibrated for (MI355X_MICROARCH.md: "On gfx950 FETCH_SIZE reports exactly 1/2 of the
// bytes of a wide coalesced streaming read ... Other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern before trusting an absolute").  The off-target look-up (crp_offtarget.hip) is a random 16-byte
// gather, its partition kernels scatter 2- and 4-byte entries: this program runs those patterns with KNOWN numbers of
// requests, lines and bytes, once under `--pmc FETCH_SIZE`, once under `--pmc WRITE_SIZE`, once under `--kernel-trace
// --stats` (tools/pmc_calibrate.sh), and tools/pmc_calibrate.py turns the three into bytes-per-request figures.
//
//   stream_read16      16 B per lane, contiguous, 2 GiB                        (the calibrated case: raw x 2 = bytes)
//   gather_stride<S>   lane i reads 16 B at byte offset i * S, S = 32 .. 256   (every 128-byte line touched 4, 2, 1, 1/2 times)
//   gather_random      52.4 M random 16-byte reads from a 256 MiB table        (the look-up's pattern)
//   stream_write16     16 B per lane, contiguous, 2 GiB
//   scatter_stride<S,W> lane i writes W = 4 or 16 bytes at byte offset i * S
//   scatter_random4    52.4 M random 4-byte writes into 0.84 GB
// Time tells what the counters cannot: a pattern whose lines-touched x 128 B / time exceeds what HBM can deliver does
// not fetch whole lines.
// Build: hipcc -O3 --offload-arch=gfx950 fetch_calibration.hip -o fetch_cal
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e__ = (x);                                                          \
        if (e__ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                   \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr uint64_t BUF_BYTES = 2ull << 30;
constexpr uint64_t TABLE_BYTES = 256ull << 20;
constexpr uint64_t N_RANDOM = 52446689;  // kept hits of the bench genome

__global__ void fill_random(uint32_t *a, uint64_t n, uint32_t mod, uint32_t seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 31;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 29;
        a[i] = (uint32_t)(x % mod);
    }
}

// (loads must not be optimised away: fold what was read and store it under a condition that never holds)
__device__ __forceinline__ void sink(uint4 v, uint32_t *out)
{
    const uint32_t f = v.x ^ v.y ^ v.z ^ v.w;
    if (f == 0x9e3779b9u) out[threadIdx.x] = f;
}

__global__ __launch_bounds__(256) void stream_read16(const uint4 *__restrict__ buf, uint64_t n, uint32_t *out)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = buf[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    sink(acc, out);
}

template <int STRIDE>
__global__ __launch_bounds__(256) void gather_stride(const uint8_t *__restrict__ buf, uint64_t n, uint32_t *out)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4 *>(buf + i * STRIDE);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    sink(acc, out);
}

__global__ __launch_bounds__(256) void gather_random(const uint32_t *__restrict__ idx, uint64_t n, const uint4 *__restrict__ table,
                                                     uint32_t *out)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = table[idx[i]];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    sink(acc, out);
}

__global__ __launch_bounds__(256) void stream_write16(uint4 *__restrict__ buf, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        buf[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

template <int STRIDE, int W>
__global__ __launch_bounds__(256) void scatter_stride(uint8_t *__restrict__ buf, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (W == 16) *reinterpret_cast<uint4 *>(buf + i * STRIDE) = make_uint4((uint32_t)i, 1, 2, 3);
        else *reinterpret_cast<uint32_t *>(buf + i * STRIDE) = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void scatter_random4(const uint32_t *__restrict__ idx, uint64_t n, uint32_t *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[idx[i]] = (uint32_t)i;
}

template <class F>
static void timed(const char *name, double useful_bytes, double requests, double lines, F launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();  // warm-up (also: the profiler sees two launches per pattern; the summary averages them)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("{\"pattern\": \"%s\", \"ms\": %.4f, \"useful_bytes\": %.0f, \"requests\": %.0f, \"lines_128B\": %.0f, "
           "\"GBs_if_whole_lines\": %.1f, \"GBs_useful\": %.1f}\n",
           name, ms, useful_bytes, requests, lines, lines * 128 / (ms * 1e-3) / 1e9, useful_bytes / (ms * 1e-3) / 1e9);
}

int main()
{
    uint8_t *buf;
    uint4 *table;
    uint32_t *idx, *idx_out, *out;
    CHECK(hipMalloc(&buf, BUF_BYTES));
    CHECK(hipMalloc(&table, TABLE_BYTES));
    CHECK(hipMalloc(&idx, N_RANDOM * 4));
    CHECK(hipMalloc(&idx_out, N_RANDOM * 4));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(buf, 1, BUF_BYTES));
    CHECK(hipMemset(table, 1, TABLE_BYTES));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, idx, N_RANDOM, (uint32_t)(TABLE_BYTES / 16), 1u);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, idx_out, N_RANDOM, (uint32_t)(N_RANDOM * 4 / 4), 2u);
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    const double B = (double)BUF_BYTES;
    timed("stream_read16", B, B / 16, B / 128, [&] { hipLaunchKernelGGL(stream_read16, grid, block, 0, 0, (const uint4 *)buf, BUF_BYTES / 16, out); });
#define GS(S) timed("gather_stride_" #S, B / S * 16, B / S, S >= 128 ? B / S : B / 128, \
                    [&] { hipLaunchKernelGGL(gather_stride<S>, grid, block, 0, 0, buf, BUF_BYTES / S, out); })
    GS(32);
    GS(64);
    GS(128);
    GS(256);
    timed("gather_random_256MiB", (double)N_RANDOM * 16, (double)N_RANDOM, (double)N_RANDOM,
          [&] { hipLaunchKernelGGL(gather_random, grid, block, 0, 0, idx, N_RANDOM, table, out); });
    timed("stream_write16", B, B / 16, B / 128, [&] { hipLaunchKernelGGL(stream_write16, grid, block, 0, 0, (uint4 *)buf, BUF_BYTES / 16); });
#define SS(S, W) timed("scatter_stride_" #S "_w" #W, B / S * W, B / S, S >= 128 ? B / S : B / 128, \
                       [&] { hipLaunchKernelGGL((scatter_stride<S, W>), grid, block, 0, 0, buf, BUF_BYTES / S); })
    SS(32, 4);
    SS(64, 4);
    SS(128, 4);
    SS(64, 16);
    SS(128, 16);
    timed("scatter_random4_0.2GB", (double)N_RANDOM * 4, (double)N_RANDOM, (double)N_RANDOM,
          [&] { hipLaunchKernelGGL(scatter_random4, grid, block, 0, 0, idx_out, N_RANDOM, (uint32_t *)buf); });
    return 0;
}
