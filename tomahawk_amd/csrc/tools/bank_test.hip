// Dev tool: do VGPR bank conflicts (reg index mod 4) slow v_and_b32 / v_bcnt_u32_b32 on gfx950?
// 128 pairs per loop iteration so that the loop branch does not dominate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#define PAIR(T,A,B,C) "v_and_b32 v" #T ", v" #A ", v" #B "\n\t" "v_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
// a: v40..47  b: v48..55  t: v56..63  c: v64..71 ; bank = index mod 4
// M0 no conflicts anywhere: and(a bank x, b bank x+1) -> t bank x+2 ; bcnt(t bank x+2, c bank x+3)
#define M0 PAIR(58,40,49,67) PAIR(59,41,50,64) PAIR(56,42,51,65) PAIR(57,43,48,66) PAIR(62,44,53,71) PAIR(63,45,54,68) PAIR(60,46,55,69) PAIR(61,47,52,70)
// M1 and sources same bank, bcnt sources distinct
#define M1 PAIR(58,40,48,67) PAIR(59,41,49,64) PAIR(56,42,50,65) PAIR(57,43,51,66) PAIR(62,44,52,71) PAIR(63,45,53,68) PAIR(60,46,54,69) PAIR(61,47,55,70)
// M2 and distinct, bcnt sources same bank (t and c same bank)
#define M2 PAIR(58,40,49,66) PAIR(59,41,50,67) PAIR(56,42,51,64) PAIR(57,43,48,65) PAIR(62,44,53,70) PAIR(63,45,54,71) PAIR(60,46,55,68) PAIR(61,47,52,69)
// M3 everything in one bank per pair
#define M3 PAIR(56,40,48,64) PAIR(57,41,49,65) PAIR(58,42,50,66) PAIR(59,43,51,67) PAIR(60,44,52,68) PAIR(61,45,53,69) PAIR(62,46,54,70) PAIR(63,47,55,71)
// M4 like M0 but the and destination shares the bank of one of its sources
#define M4 PAIR(56,40,49,67) PAIR(57,41,50,64) PAIR(58,42,51,65) PAIR(59,43,48,66) PAIR(60,44,53,71) PAIR(61,45,54,68) PAIR(62,46,55,69) PAIR(63,47,52,70)
#define R16(X) X X X X X X X X X X X X X X X X
template<int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters){
  for(int it=0; it<iters; ++it){
    if(MODE==0) asm volatile(R16(M0) ::: CLOB);
    else if(MODE==1) asm volatile(R16(M1) ::: CLOB);
    else if(MODE==2) asm volatile(R16(M2) ::: CLOB);
    else if(MODE==3) asm volatile(R16(M3) ::: CLOB);
    else asm volatile(R16(M4) ::: CLOB);
  }
  uint32_t s; asm volatile("v_add_u32 %0, v64, v65" : "=v"(s)); out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE> void run(const char* name){
  int blocks=256*2; uint32_t* d; CK(hipMalloc(&d,(size_t)blocks*256*4));
  int iters=40000; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(256),0,0,d,1000); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(256),0,0,d,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double pairs=(double)blocks*256*iters*128;
  printf("%-40s %.3f ms  word-pairs/s %.3e\n",name,ms,pairs/ms*1e3); fflush(stdout);
}
int main(){ run<0>("no bank conflict"); run<1>("and srcs same bank"); run<2>("bcnt srcs same bank"); run<3>("all same bank"); run<4>("no src conflict, and dst=src bank"); return 0; }
