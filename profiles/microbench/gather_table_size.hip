// gather_table_size.hip -- the ONE variable of the off-target look-up that had not been measured:
// how fast 52.4 M random gathers run as a function of the TABLE'S FOOTPRINT and of the ENTRY SIZE.  The look-up
// (crp_offtarget.hip, ot_lookup_kernel) reads one 16-byte entry {sites at distance 0, 1, 2, 3} per kept hit out of a
// 4^12-entry table: 256 MiB, exactly the size of the Infinity Cache.  An 8-byte entry (4 x u16, saturating, with a side
// table for the few seeds above 65 535) would halve the footprint to 128 MiB.
//
//   gather<16>  table of 16-byte entries, sizes 16 MiB .. 512 MiB      (256 MiB = today's look-up)
//   gather<8>   table of  8-byte entries, sizes  8 MiB .. 256 MiB      (128 MiB = the 4^12 x 8 B table)
//   scatter16   52.4 M 16-byte stores to RANDOM rows of a 52.4 M-row table (0.84 GB), values streamed in: the dominant
//               step of the seed-ordered alternative (look the counts up in seed order out of an
//               LDS slice of the table, then put every result where its hit lives) -- that variant cannot be faster
//               than this store pattern alone
// Indices: uniform random (the worst case; real seeds repeat).  Each pattern runs once to warm up and once timed; under
// `rocprofv3 --pmc FETCH_SIZE` the same binary gives the bytes the fabric delivered per request.
// Build: hipcc -O3 --offload-arch=gfx950 gather_table_size.hip -o gather_table_size
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

// one 16-byte store per request, like the look-up's (4 x u32 counts per hit)
template <int ENTRY>
__global__ __launch_bounds__(256) void gather(const uint32_t *__restrict__ idx, uint64_t n, const uint8_t *__restrict__ table,
                                              uint32_t mask, uint4 *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = idx[i] & mask;
    uint4 v;
    if (ENTRY == 16) {
        v = *reinterpret_cast<const uint4 *>(table + (uint64_t)k * 16);
    } else {
        const uint2 p = *reinterpret_cast<const uint2 *>(table + (uint64_t)k * 8);
        v = make_uint4(p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16);
    }
    __builtin_nontemporal_store(v.x, &out[i].x);
    __builtin_nontemporal_store(v.y, &out[i].y);
    __builtin_nontemporal_store(v.z, &out[i].z);
    __builtin_nontemporal_store(v.w, &out[i].w);
}

__global__ __launch_bounds__(256) void scatter16(const uint32_t *__restrict__ idx, uint64_t n, uint32_t rows, uint4 *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = idx[i] % rows;
    out[k] = make_uint4((uint32_t)i, k, 2, 3);
}

template <int ENTRY>
static void run(const uint32_t *idx, const uint8_t *table, uint4 *out, uint64_t table_bytes)
{
    const uint32_t entries = (uint32_t)(table_bytes / ENTRY);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const dim3 grid((uint32_t)((N_RANDOM + 255) / 256)), block(256);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(gather<ENTRY>, grid, block, 0, 0, idx, N_RANDOM, table, entries - 1, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    printf("{\"pattern\": \"gather%d_%lluMiB\", \"entry_bytes\": %d, \"table_MiB\": %llu, \"ms\": %.4f, \"requests\": %llu, "
           "\"Grequests_per_s\": %.2f, \"GBs_if_whole_lines\": %.1f, \"GBs_streams_only\": %.1f}\n",
           ENTRY, (unsigned long long)(table_bytes >> 20), ENTRY, (unsigned long long)(table_bytes >> 20), best,
           (unsigned long long)N_RANDOM, N_RANDOM / (best * 1e-3) / 1e9, N_RANDOM * 128.0 / (best * 1e-3) / 1e9,
           N_RANDOM * 20.0 / (best * 1e-3) / 1e9);
}

int main()
{
    uint8_t *table;
    uint32_t *idx;
    uint4 *out;
    CHECK(hipMalloc(&table, 512ull << 20));
    CHECK(hipMalloc(&idx, N_RANDOM * 4));
    CHECK(hipMalloc(&out, N_RANDOM * 16));
    CHECK(hipMemset(table, 1, 512ull << 20));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, idx, N_RANDOM, 0xffffffffu, 1u);
    CHECK(hipDeviceSynchronize());
    for (uint64_t mib : {16, 32, 64, 128, 256, 512}) run<16>(idx, table, out, mib << 20);
    for (uint64_t mib : {8, 16, 32, 64, 128, 256}) run<8>(idx, table, out, mib << 20);
    {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(scatter16, dim3((uint32_t)((N_RANDOM + 255) / 256)), dim3(256), 0, 0, idx, N_RANDOM, (uint32_t)N_RANDOM, out);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        printf("{\"pattern\": \"scatter16_random_800MiB\", \"entry_bytes\": 16, \"table_MiB\": %llu, \"ms\": %.4f, \"requests\": %llu, "
               "\"Grequests_per_s\": %.2f}\n", (unsigned long long)((N_RANDOM * 16) >> 20), best, (unsigned long long)N_RANDOM,
               N_RANDOM / (best * 1e-3) / 1e9);
    }
    return 0;
}
