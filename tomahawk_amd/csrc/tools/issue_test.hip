// Dev tool: how the grouping of v_and / v_bcnt and interleaved scalar instructions affects issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#define A8(B) "v_and_b32 v56, v40, v" #B "\n\tv_and_b32 v57, v41, v" #B "\n\tv_and_b32 v58, v42, v" #B "\n\tv_and_b32 v59, v43, v" #B "\n\tv_and_b32 v60, v44, v" #B "\n\tv_and_b32 v61, v45, v" #B "\n\tv_and_b32 v62, v46, v" #B "\n\tv_and_b32 v63, v47, v" #B "\n\t"
#define C8 "v_bcnt_u32_b32 v64, v56, v64\n\tv_bcnt_u32_b32 v65, v57, v65\n\tv_bcnt_u32_b32 v66, v58, v66\n\tv_bcnt_u32_b32 v67, v59, v67\n\tv_bcnt_u32_b32 v68, v60, v68\n\tv_bcnt_u32_b32 v69, v61, v69\n\tv_bcnt_u32_b32 v70, v62, v70\n\tv_bcnt_u32_b32 v71, v63, v71\n\t"
#define P(T,A,B,C) "v_and_b32 v" #T ", v" #A ", v" #B "\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define I8(B) P(56,40,B,64) P(57,41,B,65) P(58,42,B,66) P(59,43,B,67) P(60,44,B,68) P(61,45,B,69) P(62,46,B,70) P(63,47,B,71)
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
#define G4(X) X(48) X(49) X(50) X(51)
#define G16(X) G4(X) G4(X) G4(X) G4(X)
#define GRP(B) A8(B) C8
#define GRPN(B) A8(B) C8 "s_nop 0\n\t"
#define GRPS(B) A8(B) C8 "s_add_u32 s20, s20, 1\n\t"
#define GRPA(B) A8(B) "s_nop 0\n\t" C8
template<int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters){
  for(int it=0; it<iters; ++it){
    if(MODE==0)      asm volatile(G16(GRP)  ::: CLOB);
    else if(MODE==1) asm volatile(G16(GRPN) ::: CLOB);
    else if(MODE==2) asm volatile(G16(I8)   ::: CLOB);
    else if(MODE==3) asm volatile(G16(GRPS) ::: CLOB, "s20");
    else if(MODE==4) asm volatile(G16(GRPA) ::: CLOB);
  }
  uint32_t s; asm volatile("v_add_u32 %0, v64, v65" : "=v"(s)); out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE> void run(const char* name, int bpc){
  int blocks=256*bpc; uint32_t* d; CK(hipMalloc(&d,(size_t)blocks*256*4));
  int iters=40000; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(256),0,0,d,1000); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(256),0,0,d,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double pairs=(double)blocks*256*iters*128;
  printf("%-34s waves/SIMD=%d  %.3f ms  word-pairs/s %.3e\n",name,bpc,ms,pairs/ms*1e3); fflush(stdout); CK(hipFree(d));
}
int main(){ for(int b: {2,4}){ run<0>("8and+8bcnt groups",b); run<1>("groups + s_nop 0",b); run<2>("interleaved and/bcnt",b); run<4>("8and,s_nop,8bcnt",b);} return 0; }
