// Dev tool: issue rate of v_and_b32 / v_bcnt_u32_b32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int MODE,int UNR> __global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters){
  uint32_t acc[32], x[8];
  for(int i=0;i<32;++i) acc[i]=threadIdx.x+i;
  for(int i=0;i<8;++i) x[i]=seed*(threadIdx.x+i+1);
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int r=0;r<UNR;++r){
#pragma unroll
      for(int i=0;i<32;++i){
        if(MODE==0){ asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[i&7])); }
        else if(MODE==1){ asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[i&7])); }
        else if(MODE==2){ uint32_t t; asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "v"(x[i&7]), "v"(x[(i+r+1)&7])); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(t)); }
        else if(MODE==3){ asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[i&7])); }
      }
    }
  }
  uint32_t s=0; for(int i=0;i<32;++i) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE,int UNR> void run(const char* name, int wpc){
  int blocks=256*wpc/4*1; // wpc waves per CU -> blocks of 4 waves
  uint32_t* d; CK(hipMalloc(&d, (size_t)blocks*256*4));
  int iters=80000/UNR; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE,UNR>),dim3(blocks),dim3(256),0,0,d,3u,100); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE,UNR>),dim3(blocks),dim3(256),0,0,d,3u,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double instr = (double)blocks*256*iters*32*UNR*(MODE==2?2:1);
  printf("%-10s waves/CU=%2d  %.3f ms  lane-ops/s %.3e  (%.1f%% of 7.86e13)\n",name,wpc,ms,instr/ms*1e3, instr/ms*1e3/7.864e13*100);
  CK(hipFree(d));
}
int main(){
  for(int wpc : {8}){ run<2,4>("ab u4",wpc); run<2,16>("ab u16",wpc); run<2,64>("ab u64",wpc); run<2,128>("ab u128",wpc); run<0,64>("bcnt u64",wpc); run<1,64>("and u64",wpc);}
  return 0;
}
