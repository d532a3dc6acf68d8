// Dev tool (round 6): would the three-product loop run faster with its A operands in SGPRs?  A wave = 8 A variants (their H / Q words by s_load_dwordx2
// from global memory, the carriers H | Q made in place by s_or_b32), a lane = one B variant (its words by ds_read_b64 from LDS): 2 LDS reads and 16
// scalar loads per 48 products where the kernel has 8 LDS reads.  Streams with every register fixed by hand (sgpr_probe_gen.py), 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "sgpr_probe_gen.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#define VCLOB "v10","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v32","v33","v34","v35","v36","v37","v38","v39"
#define SCLOB "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51", \
              "s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83", \
              "s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99","s100","s101"
constexpr uint32_t PITCH = 65536 + 128;
template<int MODE> __global__ __launch_bounds__(512, 2) void k(const uint32_t* __restrict__ rows, uint32_t* out, int iters){
  __shared__ uint32_t lds[16384];
  for(int i=threadIdx.x;i<16384;i+=512) lds[i]=i; __syncthreads();
  const uint32_t b=(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x&63)*256 + ((threadIdx.x&7)<<4);
  asm volatile("v_mov_b32 v10, %0" :: "v"(b) : "v10");
  asm volatile("s_mov_b32 s84, 0\n\ts_mov_b32 s85, %0\n\ts_mov_b32 s86, %1\n\ts_mov_b32 s87, %2\n\ts_mov_b32 s88, %3\n\ts_mov_b32 s89, %4\n\ts_mov_b32 s90, %5\n\ts_mov_b32 s91, %6\n\t"
               "s_mov_b32 s92, %7\n\ts_mov_b32 s93, %8\n\ts_mov_b32 s94, %9\n\ts_mov_b32 s95, %10\n\ts_mov_b32 s96, %11\n\ts_mov_b32 s97, %12\n\ts_mov_b32 s98, %13\n\ts_mov_b32 s99, %14"
               :: "n"(PITCH), "n"(2*PITCH), "n"(3*PITCH), "n"(4*PITCH), "n"(5*PITCH), "n"(6*PITCH), "n"(7*PITCH), "n"(8*PITCH), "n"(9*PITCH), "n"(10*PITCH), "n"(11*PITCH), "n"(12*PITCH), "n"(13*PITCH), "n"(14*PITCH), "n"(15*PITCH) : SCLOB);
  const uint32_t* base = rows + (size_t)(blockIdx.x & 63) * 16 * (PITCH / 4);      // 64 different row groups over the chip
  for(int it=0; it<iters; ++it){
    const uint32_t* p = base + (it & 511) * 32;      // 8 half-slots of 2 words per iteration, and on along the rows
    asm volatile("s_mov_b64 s[100:101], %0" :: "s"(p) : "s100", "s101");
    if(MODE==0) asm volatile(STEP_SGPR_ONLY STEP_SGPR_ONLY STEP_SGPR_ONLY STEP_SGPR_ONLY ::: VCLOB, SCLOB);
    else if(MODE==1) asm volatile(STEP_SGPR_OR STEP_SGPR_OR STEP_SGPR_OR STEP_SGPR_OR ::: VCLOB, SCLOB, "scc");
    else asm volatile(STEP_SGPR_FULL STEP_SGPR_FULL STEP_SGPR_FULL STEP_SGPR_FULL ::: VCLOB, SCLOB, "scc", "memory");
  }
  uint32_t s; asm volatile("v_add_u32 %0, v12, v20" : "=v"(s)); out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE> void run(const char* name, const uint32_t* rows){
  int blocks=256*2; uint32_t* d; CK(hipMalloc(&d,(size_t)blocks*512*4));
  int iters=20000; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,rows,d,2000); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,rows,d,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  const double products=(double)blocks*512*iters*8*48;      // 8 half-slots of 48 products per iteration
  printf("%-70s %.3f ms  products/s %.3e (%.1f %% of the and+bcnt ceiling 2.62e13)\n",name,ms,products/ms*1e3,products/ms*1e3/2.6214e13*100); fflush(stdout);
}
int main(){
  uint32_t* rows; const size_t bytes=(size_t)64*16*PITCH + (1<<20); CK(hipMalloc(&rows,bytes)); CK(hipMemset(rows,0x5a,bytes));
  run<0>("products with their A operand in an SGPR, nothing else",rows);
  run<1>("... + the 16 s_or_b32 per half-slot that make the carriers in place",rows);
  run<2>("... + 16 s_load_dwordx2 and 2 ds_read_b64 per half-slot (48 products)",rows);
  return 0;
}
