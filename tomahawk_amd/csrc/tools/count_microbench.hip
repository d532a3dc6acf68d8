// Stand-alone micro-benchmark + self-check of the count kernel (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../hip/ld_count.hip.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s @%d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)
int main(int argc,char**argv){
  uint32_t R = argc>1? atoi(argv[1]) : 4096;     // rows (multiple of 128)
  uint32_t W = argc>2? atoi(argv[2]) : 3136;     // words per row (multiple of 32)
  int reps = argc>3? atoi(argv[3]) : 3;
  size_t nw=(size_t)R*W;
  std::vector<uint32_t> h(nw); std::mt19937 rng(1); for(auto&x:h) x=rng();
  uint32_t *d,*C; CK(hipMalloc(&d,nw*4)); CK(hipMalloc(&C,(size_t)R*R*4));
  CK(hipMemcpy(d,h.data(),nw*4,hipMemcpyHostToDevice)); CK(hipMemset(C,0xff,(size_t)R*R*4));
  dim3 grid(R/128,R/128), block(twk::COUNT_THREADS);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int diag=0; diag<2; ++diag){
    hipLaunchKernelGGL((twk::k_count_tile_t<twk::COUNT_NW>),grid,block,0,0,d,W,0u,0u,diag,C,R); CK(hipDeviceSynchronize());
    float best=1e30f;
    for(int i=0;i<reps;++i){ CK(hipEventRecord(e0)); hipLaunchKernelGGL((twk::k_count_tile_t<twk::COUNT_NW>),grid,block,0,0,d,W,0u,0u,diag,C,R); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
    double tiles = diag? (double)(R/128)*(R/128+1)/2 : (double)(R/128)*(R/128);
    double rowpairs = tiles*128*128;
    double wordops = rowpairs*W;            // and+bcnt pairs
    printf("diag=%d R=%u W=%u best %.3f ms  rowpairs/s %.3e  word-pairs/s %.3e  VALU lane-ops/s %.3e (peak 7.86e13)\n",diag,R,W,best,rowpairs/best*1e3,wordops/best*1e3,2*wordops/best*1e3);
  }
  // check (diag run left lower tiles stale from full run: fine, both valid)
  std::vector<uint32_t> hc((size_t)R*R); CK(hipMemcpy(hc.data(),C,(size_t)R*R*4,hipMemcpyDeviceToHost));
  int bad=0; std::mt19937 r2(7);
  for(int s=0;s<2000;++s){ uint32_t i=r2()%R,j=r2()%R; uint32_t ref=0; for(uint32_t k=0;k<W;++k) ref+=__builtin_popcount(h[(size_t)i*W+k]&h[(size_t)j*W+k]); if(ref!=hc[(size_t)i*R+j]){ if(bad<5) printf("MISMATCH (%u,%u) ref %u got %u\n",i,j,ref,hc[(size_t)i*R+j]); ++bad; } }
  printf("check: %d mismatches of 2000\n",bad);
  return bad!=0;
}
