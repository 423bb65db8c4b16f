// gather_scatter16.hip -- what a random 16-byte GATHER and a random 16-byte SCATTER cost on MI355X.
//
// Prices the two ways the off-target look-up (crp_offtarget.hip) could be organised:
//   gather   out[i] = table[seed[i]]      hits in genome order, 16.7 M x 16 B table (what ships)
//   scatter  out[perm[i]] = value(i)      hits in seed order (table rows through LDS), counts scattered
//                                         back to the hits' own slots
// n = 52.4 M items as on the bench genome.  Build: hipcc -O3 --offload-arch=gfx950 gather_scatter16.hip -o gs16
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

// a permutation of [0, n): multiplication by an odd constant modulo the next power of two, cycle-walked into range
__global__ void fill_perm(uint32_t *a, uint64_t n, uint32_t bits)
{
    const uint64_t mask = (1ull << bits) - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i;
        do {
            x = (x * 0x2545F491ull + 0x1234567ull) & mask;
            x ^= x >> (bits / 2);  // (an involution-free mix is not needed: only the spread matters)
            x = (x * 0x9E3779B1ull) & mask;
        } while (x >= n);
        a[i] = (uint32_t)x;
    }
}

__global__ __launch_bounds__(256) void gather16(const uint32_t *__restrict__ idx, uint64_t n, const uint4 *__restrict__ table,
                                                uint4 *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = table[idx[i]];
}

// the same gather from a table of 8-byte entries (four 16-bit counts), widened to 16 bytes on the way out
__global__ __launch_bounds__(256) void gather8(const uint32_t *__restrict__ idx, uint64_t n, const uint2 *__restrict__ table,
                                               uint4 *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint2 v = table[idx[i]];
        out[i] = make_uint4(v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16);
    }
}

// ... and from 4-byte entries
__global__ __launch_bounds__(256) void gather4(const uint32_t *__restrict__ idx, uint64_t n, const uint32_t *__restrict__ table,
                                               uint4 *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t v = table[idx[i]];
        out[i] = make_uint4(v & 0xffu, (v >> 8) & 0xffu, (v >> 16) & 0xffu, v >> 24);
    }
}

__global__ __launch_bounds__(256) void scatter16(const uint32_t *__restrict__ dst, uint64_t n, uint4 *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t v = (uint32_t)i;
        out[dst[i]] = make_uint4(v, v + 1, v + 2, v + 3);
    }
}

__global__ __launch_bounds__(256) void scatter16_nt(const uint32_t *__restrict__ dst, uint64_t n, uint4 *__restrict__ out)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t v = (uint32_t)i;
        u32x4 t = {v, v + 1, v + 2, v + 3};
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(out + dst[i]));
    }
}

template <class F>
static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();  // clocks up
    CHECK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const uint64_t n = 52446689;        // hits of the bench genome
    const uint32_t table_n = 1u << 24;  // 4^12 seeds
    uint32_t *idx, *perm;
    uint4 *table, *out;
    CHECK(hipMalloc(&idx, n * 4));
    CHECK(hipMalloc(&perm, n * 4));
    CHECK(hipMalloc(&table, (size_t)table_n * 16));
    CHECK(hipMalloc(&out, n * 16));
    CHECK(hipMemset(table, 1, (size_t)table_n * 16));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, idx, n, table_n, 7u);
    hipLaunchKernelGGL(fill_perm, dim3(4096), dim3(256), 0, 0, perm, n, 26u);
    CHECK(hipDeviceSynchronize());
    const int grid = 16384;
    const float g = time_ms([&] { hipLaunchKernelGGL(gather16, dim3(grid), dim3(256), 0, 0, idx, n, table, out); }, 20);
    const float g8 = time_ms(
        [&] { hipLaunchKernelGGL(gather8, dim3(grid), dim3(256), 0, 0, idx, n, reinterpret_cast<const uint2 *>(table), out); }, 20);
    const float g4 = time_ms(
        [&] { hipLaunchKernelGGL(gather4, dim3(grid), dim3(256), 0, 0, idx, n, reinterpret_cast<const uint32_t *>(table), out); }, 20);
    const float s = time_ms([&] { hipLaunchKernelGGL(scatter16, dim3(grid), dim3(256), 0, 0, perm, n, out); }, 20);
    const float t = time_ms([&] { hipLaunchKernelGGL(scatter16_nt, dim3(grid), dim3(256), 0, 0, perm, n, out); }, 20);
    printf("{\"items\": %llu, \"gather16_ms\": %.4f, \"gather8_ms\": %.4f, \"gather4_ms\": %.4f, \"scatter16_ms\": %.4f, "
           "\"scatter16_nontemporal_ms\": %.4f}\n",
           (unsigned long long)n, g, g8, g4, s, t);
    return 0;
}
