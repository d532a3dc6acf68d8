// Dev tool: where do the six v_or of the three-product form's 24 products per word go?  Register-only streams of
// 24 x (v_and, s_nop, v_bcnt-accumulate) with six v_or_b32 placed in different ways, 4 waves per SIMD (512-thread blocks,
// two per CU, like the count kernel).  Reference: the same 24 products without any v_or.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
// product: v_and t, a, b ; s_nop ; v_bcnt acc, t, acc     (temps alternate v56 / v57)
#define P(T,A,B,C) "v_and_b32 v" #T ", v" #A ", v" #B "\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
// the same with a v_or in the slot of the s_nop
#define PO(T,A,B,C,OD,O1,O2) "v_and_b32 v" #T ", v" #A ", v" #B "\n\tv_or_b32 v" #OD ", v" #O1 ", v" #O2 "\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define OR(D,A,B) "v_or_b32 v" #D ", v" #A ", v" #B "\n\t"
#define NOP "s_nop 0\n\t"
// 12 products against one B variant: accumulators v64..v67 (HH), v68..v71 (S); A words v33..v40, carriers v48..v51, B words v41 v42, cB v52
#define P12 P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) P(57,51,42,71)
#define OR4 OR(48,33,34) OR(49,35,36) OR(50,37,38) OR(51,39,40)
// A: no ORs
#define SA P12 P12
// B: clumped (the kernel's first form): 4 ORs, then (1 OR, 12 products) x 2
#define SB OR4 OR(52,41,42) P12 OR(52,43,44) P12
// C: spread: one OR behind each of the first bcnts
#define P12C1 P(56,33,41,64) OR(48,33,34) P(57,35,41,65) OR(49,35,36) P(56,37,41,66) OR(50,37,38) P(57,39,41,67) OR(51,39,40) P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) P(57,51,42,71)
#define SC OR(52,41,42) P12C1 OR(52,43,44) P12
// D: the OR takes the place of the s_nop in six products
#define P12D1 PO(56,33,41,64,48,33,34) PO(57,35,41,65,49,35,36) PO(56,37,41,66,50,37,38) PO(57,39,41,67,51,39,40) P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) PO(57,51,42,71,53,43,44)
#define P12D2 P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) P(56,34,53,68) P(57,36,53,69) P(56,38,53,70) P(57,40,53,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) PO(57,51,42,71,52,41,42)
#define SD P12D1 P12D2
// E: each OR with an s_nop in front of it, spread
#define P12E1 P(56,33,41,64) NOP OR(48,33,34) P(57,35,41,65) NOP OR(49,35,36) P(56,37,41,66) NOP OR(50,37,38) P(57,39,41,67) NOP OR(51,39,40) P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) P(57,51,42,71)
#define SE NOP OR(52,41,42) P12E1 NOP OR(52,43,44) P12
// F: OR, s_nop behind it, spread
#define P12F1 P(56,33,41,64) OR(48,33,34) NOP P(57,35,41,65) OR(49,35,36) NOP P(56,37,41,66) OR(50,37,38) NOP P(57,39,41,67) OR(51,39,40) NOP P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) P(57,51,42,71)
#define SF OR(52,41,42) NOP P12F1 OR(52,43,44) NOP P12
// G: all six ORs in front
#define SG OR4 OR(52,41,42) OR(53,43,44) P12 P12D2N
#define P12D2N P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) P(56,34,53,68) P(57,36,53,69) P(56,38,53,70) P(57,40,53,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) P(57,51,42,71)
// I: the OR directly behind an AND, in front of the s_nop: (and, or, nop, bcnt)
#define PI(T,A,B,C,OD,O1,O2) "v_and_b32 v" #T ", v" #A ", v" #B "\n\tv_or_b32 v" #OD ", v" #O1 ", v" #O2 "\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define P12I1 PI(56,33,41,64,48,33,34) PI(57,35,41,65,49,35,36) PI(56,37,41,66,50,37,38) PI(57,39,41,67,51,39,40) P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) PI(57,51,42,71,53,43,44)
#define P12I2 P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) P(56,34,53,68) P(57,36,53,69) P(56,38,53,70) P(57,40,53,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) PI(57,51,42,71,52,41,42)
#define SI P12I1 P12I2
// K: ORs in pairs behind a bcnt
#define P12K1 P(56,33,41,64) OR(48,33,34) OR(49,35,36) P(57,35,41,65) P(56,37,41,66) OR(50,37,38) OR(51,39,40) P(57,39,41,67) P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) P(57,51,42,71) OR(52,41,42) OR(53,43,44)
#define SK P12K1 P12D2N
// M: (or, and, nop, bcnt): the OR directly in front of an AND
#define PM(T,A,B,C,OD,O1,O2) "v_or_b32 v" #OD ", v" #O1 ", v" #O2 "\n\tv_and_b32 v" #T ", v" #A ", v" #B "\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define P12M1 PM(56,33,41,64,48,33,34) PM(57,35,41,65,49,35,36) PM(56,37,41,66,50,37,38) PM(57,39,41,67,51,39,40) P(56,34,52,68) P(57,36,52,69) P(56,38,52,70) P(57,40,52,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) PM(57,51,42,71,53,43,44)
#define P12M2 P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) P(56,34,53,68) P(57,36,53,69) P(56,38,53,70) P(57,40,53,71) P(56,48,42,68) P(57,49,42,69) P(56,50,42,70) PM(57,51,42,71,52,41,42)
#define SM P12M1 P12M2
#define R8(X) X X X X X X X X
template<int MODE> __global__ __launch_bounds__(512, 2) void k(uint32_t* out, int iters){
  for(int it=0; it<iters; ++it){
    if(MODE==0) asm volatile(R8(SA) ::: CLOB);
    else if(MODE==1) asm volatile(R8(SB) ::: CLOB);
    else if(MODE==2) asm volatile(R8(SC) ::: CLOB);
    else if(MODE==3) asm volatile(R8(SD) ::: CLOB);
    else if(MODE==4) asm volatile(R8(SE) ::: CLOB);
    else if(MODE==5) asm volatile(R8(SF) ::: CLOB);
    else if(MODE==6) asm volatile(R8(SG) ::: CLOB);
    else if(MODE==7) asm volatile(R8(SI) ::: CLOB);
    else if(MODE==8) asm volatile(R8(SK) ::: CLOB);
    else asm volatile(R8(SM) ::: CLOB);
  }
  uint32_t s; asm volatile("v_add_u32 %0, v64, v65" : "=v"(s)); out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE> void run(const char* name, int ors){
  int blocks=256*2; uint32_t* d; CK(hipMalloc(&d,(size_t)blocks*512*4));
  int iters=20000; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,d,2000); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,d,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  const double words=(double)blocks*512*iters*8;            // (lane, word) steps of 24 products
  const double cyc=ms*1e-3*2.4e9*256*4*32/words/32;          // SIMD cycles per wave64 word step at 2.4 GHz... per lane-step x 64 lanes / 32
  printf("%-52s %.3f ms  products/s %.3e  cycles per word step (24 products%s) at 2.4 GHz: %.1f\n",name,ms,words*24/ms*1e3,ors?" + 6 v_or":"",cyc*2); fflush(stdout);
}
int main(){ run<0>("A: 24 products, no v_or",0); run<1>("B: 4 v_or, then (v_or, 12 products) x 2",1); run<2>("C: v_or behind a bcnt, spread",1); run<3>("D: v_or in the place of the s_nop",1); run<4>("E: (s_nop, v_or) spread",1); run<5>("F: (v_or, s_nop) spread",1);
  run<6>("G: six v_or, then 24 products",1); run<7>("I: (and, or, nop, bcnt)",1); run<8>("K: v_or in pairs behind a bcnt",1); run<9>("M: (or, and, nop, bcnt)",1); return 0; }
