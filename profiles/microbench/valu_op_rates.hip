#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define OP4(name, fmt) \
  __global__ __launch_bounds__(256) void k_##name(unsigned *out, int iters){ \
    unsigned a0=threadIdx.x,a1=a0*3+1,a2=a0*5+2,a3=a0*7+3; \
    for(int i=0;i<iters;++i){ REP64(asm volatile(fmt : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3));) } \
    if(a0+a1+a2+a3==0x7fffffff) out[0]=1; }
OP4(and,   "v_and_b32 %0, 0x12345678, %0\n v_and_b32 %1, 0x12345678, %1\n v_and_b32 %2, 0x12345678, %2\n v_and_b32 %3, 0x12345678, %3")
OP4(lshl,  "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3")
OP4(lshr,  "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3")
OP4(mul24, "v_mul_u32_u24 %0, 512, %0\n v_mul_u32_u24 %1, 512, %1\n v_mul_u32_u24 %2, 512, %2\n v_mul_u32_u24 %3, 512, %3")
OP4(add,   "v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3")
OP4(xor,   "v_xor_b32 %0, 0x1234567, %0\n v_xor_b32 %1, 0x1234567, %1\n v_xor_b32 %2, 0x1234567, %2\n v_xor_b32 %3, 0x1234567, %3")
OP4(bfe,   "v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 1, 31")
OP4(mov,   "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0")
OP4(alignbit, "v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %0, 7")
OP4(andvv, "v_and_b32 %0, %1, %0\n v_and_b32 %1, %2, %1\n v_and_b32 %2, %3, %2\n v_and_b32 %3, %0, %3")
OP4(lshl_or, "v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %0")
OP4(bfrev, "v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3")
OP4(sdwa,  "v_mov_b32_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_mov_b32_sdwa %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_mov_b32_sdwa %2, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_mov_b32_sdwa %3, %3 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0")
template <typename K> void run(const char* name, K kern){
  unsigned *d; hipMalloc(&d, 4); const int iters=200, grid=256*8;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid,256>>>(d,10); hipDeviceSynchronize();
  hipEventRecord(e0); kern<<<grid,256>>>(d,iters); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double instr_per_simd = 8.0*iters*256; // 8 waves/SIMD
  printf("%-10s %.3f ms  -> %.2f ns/instr/SIMD\n", name, ms, ms*1e6/instr_per_simd);
  hipFree(d);
}
int main(){
  run("and_lit", k_and); run("and_vv", k_andvv); run("xor_lit", k_xor); run("add", k_add); run("mov", k_mov);
  run("lshl", k_lshl); run("lshr", k_lshr); run("mul_u24", k_mul24); run("bfe(VOP3)", k_bfe); run("alignbit", k_alignbit);
  run("lshl_or", k_lshl_or); run("bfrev", k_bfrev); run("sdwa_mov", k_sdwa);
  run("and_lit", k_and);
}
