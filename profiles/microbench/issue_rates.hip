// Issue-rate microbenchmarks for gfx950: how many cycles does a wave64 instruction cost
// per SIMD, by type and mix, at a given number of waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters, double seed)
{
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    double d0 = seed + a0, d1 = seed * 2 + a0, d2 = seed * 3, d3 = seed * 4;
    double g = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 256 independent-ish v_and_b32 (4 chains)
            REP64(asm volatile("v_and_b32 %0, 0x12345678, %0\n v_and_b32 %1, 0x12345678, %1\n v_and_b32 %2, 0x12345678, %2\n v_and_b32 %3, 0x12345678, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 1) {  // 256 v_fma_f64, 4 chains
            REP64(asm volatile("v_fma_f64 %0, %4, %4, %0\n v_fma_f64 %1, %4, %4, %1\n v_fma_f64 %2, %4, %4, %2\n v_fma_f64 %3, %4, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(g));)
        } else if (MODE == 2) {  // pairs and + fma (128 + 128)
            REP64(asm volatile("v_and_b32 %0, 0x12345678, %0\n v_fma_f64 %2, %4, %4, %2\n v_and_b32 %1, 0x12345678, %1\n v_fma_f64 %3, %4, %4, %3" : "+v"(a0), "+v"(a1), "+v"(d0), "+v"(d1) : "v"(g));)
        } else if (MODE == 3) {  // the kernel's pattern: s_mov, s_mov, v_and, v_fmac with SGPR constant (64 terms x 4)
            REP64(asm volatile("s_mov_b32 s8, 0x94d0063d\n s_mov_b32 s9, 0x3f29c542\n v_and_b32 %0, 0x800000, %0\n v_fma_f64 %1, s[8:9], %2, %1" : "+v"(a0), "+v"(d0) : "v"(g) : "s8", "s9");)
        } else if (MODE == 4) {  // 256 s_mov only
            REP64(asm volatile("s_mov_b32 s8, 0x94d0063d\n s_mov_b32 s9, 0x3f29c542\n s_mov_b32 s8, 0x14d0063d\n s_mov_b32 s9, 0x2f29c542" ::: "s8", "s9");)
        } else if (MODE == 5) {  // v_and + v_fma with constants preloaded in VGPRs (no SALU): 128+128
            REP64(asm volatile("v_and_b32 %0, 0x800000, %0\n v_fma_f64 %2, %4, %4, %2\n v_and_b32 %1, 0x800000, %1\n v_fma_f64 %3, %4, %4, %3" : "+v"(a0), "+v"(a1), "+v"(d0), "+v"(d1) : "v"(g));)
        } else if (MODE == 6) {  // v_add_f64 x256
            REP64(asm volatile("v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(g));)
        } else if (MODE == 7) {  // v_lshlrev_b32 x256
            REP64(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 8) {  // v_bitop3 x256 (VOP3)
            REP64(asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a0 + a1 + a2 + a3 == 0x7fffffff && d0 + d1 + d2 + d3 == 1.25) out[0] = 0;
}

template <int MODE>
void run(const char *name, int blocks_per_cu)
{
    const int n_cu = 256, iters = 200;
    const int grid = n_cu * blocks_per_cu;
    unsigned long long *d;
    hipMalloc(&d, grid * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, iters, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= grid;
    // each wave issues iters*256 instructions; waves per SIMD = blocks_per_cu (4 waves per block, 4 SIMDs)
    const double instr_per_wave = (double)iters * 256;
    printf("%-34s waves/SIMD %d: %.2f memtime-ticks per instr per wave, %.2f per instr per SIMD, wall %.3f ms\n", name,
           blocks_per_cu, avg / instr_per_wave, avg / instr_per_wave / blocks_per_cu, ms);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_and_b32", w);
        run<1>("v_fma_f64", w);
        run<6>("v_add_f64", w);
        run<7>("v_lshlrev_b32", w);
        run<8>("v_and_or_b32 (VOP3)", w);
        run<2>("and+fma pairs", w);
        run<3>("s_mov,s_mov,v_and,v_fma(sgpr)", w);
        run<4>("s_mov_b32", w);
        printf("\n");
    }
    return 0;
}
