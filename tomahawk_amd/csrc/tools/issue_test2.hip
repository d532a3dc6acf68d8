// Dev tool: mutate the fast stream (and t; s_nop; bcnt acc,t,acc) step by step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
// F: fast reference: single temp v41, s_nop between
#define F(A,B,C) "v_and_b32 v41, v" #A ", v" #B "\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v41, v" #C "\n\t"
// G: same without s_nop
#define G(A,B,C) "v_and_b32 v41, v" #A ", v" #B "\n\tv_bcnt_u32_b32 v" #C ", v41, v" #C "\n\t"
// H: rotating temps v56..v63, no nop
#define H(T,A,B,C) "v_and_b32 v" #T ", v" #A ", v" #B "\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define F8 F(33,34,64) F(34,35,65) F(35,36,66) F(36,37,67) F(37,38,68) F(38,39,69) F(39,40,70) F(40,33,71)
#define G8 G(33,34,64) G(34,35,65) G(35,36,66) G(36,37,67) G(37,38,68) G(38,39,69) G(39,40,70) G(40,33,71)
#define H8 H(56,33,34,64) H(57,34,35,65) H(58,35,36,66) H(59,36,37,67) H(60,37,38,68) H(61,38,39,69) H(62,39,40,70) H(63,40,33,71)
// I: rotating temps + s_nop between and and bcnt
#define I(T,A,B,C) "v_and_b32 v" #T ", v" #A ", v" #B "\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define I8 I(56,33,34,64) I(57,34,35,65) I(58,35,36,66) I(59,36,37,67) I(60,37,38,68) I(61,38,39,69) I(62,39,40,70) I(63,40,33,71)
// J: fast reference but accumulators only 2 registers (dependency chain on acc)
#define J8 F(33,34,64) F(34,35,65) F(35,36,64) F(36,37,65) F(37,38,64) F(38,39,65) F(39,40,64) F(40,33,65)
// K: 8 ands, then (s_nop, bcnt) x 8
#define KA(T,A,B) "v_and_b32 v" #T ", v" #A ", v" #B "\n\t"
#define KB(T,C) "s_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define K8 KA(56,33,34) KA(57,34,35) KA(58,35,36) KA(59,36,37) KA(60,37,38) KA(61,38,39) KA(62,39,40) KA(63,40,33) KB(56,64) KB(57,65) KB(58,66) KB(59,67) KB(60,68) KB(61,69) KB(62,70) KB(63,71)
// L: 8 x (nop, and) then 8 x (nop, bcnt): every VALU behind a non-VALU
#define LA(T,A,B) "s_nop 0\n\tv_and_b32 v" #T ", v" #A ", v" #B "\n\t"
#define L8 LA(56,33,34) LA(57,34,35) LA(58,35,36) LA(59,36,37) LA(60,37,38) LA(61,38,39) LA(62,39,40) LA(63,40,33) KB(56,64) KB(57,65) KB(58,66) KB(59,67) KB(60,68) KB(61,69) KB(62,70) KB(63,71)
// N: and_k, nop, bcnt_{k-1} software pipelined (bcnt never depends on the instruction before it)
#define NP(T,A,B,TP,C) "v_and_b32 v" #T ", v" #A ", v" #B "\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #TP ", v" #C "\n\t"
#define N8 NP(56,33,34,63,71) NP(57,34,35,56,64) NP(58,35,36,57,65) NP(59,36,37,58,66) NP(60,37,38,59,67) NP(61,38,39,60,68) NP(62,39,40,61,69) NP(63,40,33,62,70)
// O: two ands, then nop,bcnt,nop,bcnt
#define O8 KA(56,33,34) KA(57,34,35) KB(56,64) KB(57,65) KA(58,35,36) KA(59,36,37) KB(58,66) KB(59,67) KA(60,37,38) KA(61,38,39) KB(60,68) KB(61,69) KA(62,39,40) KA(63,40,33) KB(62,70) KB(63,71)
#define R16(X) X X X X X X X X X X X X X X X X
template<int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters){
  for(int it=0; it<iters; ++it){
    if(MODE==0) asm volatile(R16(F8) ::: CLOB);
    else if(MODE==1) asm volatile(R16(G8) ::: CLOB);
    else if(MODE==2) asm volatile(R16(H8) ::: CLOB);
    else if(MODE==3) asm volatile(R16(I8) ::: CLOB);
    else if(MODE==4) asm volatile(R16(J8) ::: CLOB);
    else if(MODE==5) asm volatile(R16(K8) ::: CLOB);
    else if(MODE==6) asm volatile(R16(L8) ::: CLOB);
    else if(MODE==7) asm volatile(R16(N8) ::: CLOB);
    else asm volatile(R16(O8) ::: CLOB);
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
  printf("%-44s %.3f ms  word-pairs/s %.3e\n",name,ms,pairs/ms*1e3); fflush(stdout);
}
int main(){ run<0>("F: 1 temp, s_nop between"); run<1>("G: 1 temp, no nop"); run<2>("H: 8 temps, no nop"); run<3>("I: 8 temps, s_nop between"); run<4>("J: F with 2 accumulators"); run<5>("K: 8 and, 8x(nop,bcnt)"); run<6>("L: 8x(nop,and), 8x(nop,bcnt)"); run<7>("N: and_k,nop,bcnt_k-1"); run<8>("O: 2and,2x(nop,bcnt)"); return 0; }
