// Stand-alone micro-benchmark + self-check of the count kernels (dev tool):
//   grid  k_count_tile_t   one block per 128 x 128 tile (2-D grid; the round-1 kernel)
//   list  k_count_list_t   persistent blocks over a tile list, data-parallel rounds + stream-K tail
// usage: count_microbench [rows=4096] [words=3136] [reps=3] [blocks=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../hip/ld_count.hip.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s @%d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)

static std::vector<uint32_t> make_list(uint32_t g, int diag, uint32_t P, bool patch) {
	std::vector<uint32_t> seq;
	if (!patch) { for (uint32_t y = 0; y < g; ++y) for (uint32_t x = diag ? y : 0; x < g; ++x) seq.push_back(y << 16 | x); return seq; }
	for (uint32_t py = 0; py < g; py += 8) for (uint32_t px = 0; px < g; px += 8)
		for (uint32_t y = py; y < std::min(py + 8, g); ++y) for (uint32_t x = std::max(px, diag ? y : 0u); x < std::min(px + 8, g); ++x) seq.push_back(y << 16 | x);
	std::vector<uint32_t> out(seq.size());
	const size_t T = seq.size(), rounds = T / P, per = P / 8;
	for (size_t r = 0; r < rounds; ++r) for (size_t k = 0; k < 8; ++k) for (size_t j = 0; j < per; ++j) out[r * P + 8 * j + k] = seq[(r * 8 + k) * per + j];
	for (size_t i = rounds * P; i < T; ++i) out[i] = seq[i];
	return out;
}

int main(int argc,char**argv){
  uint32_t R = argc>1? atoi(argv[1]) : 4096;     // rows (multiple of 128)
  uint32_t W = argc>2? atoi(argv[2]) : 3136;     // words per row (multiple of 32)
  int reps = argc>3? atoi(argv[3]) : 3;
  uint32_t P = argc>4? atoi(argv[4]) : 512;      // persistent blocks
  size_t nw=(size_t)R*W;
  std::vector<uint32_t> h(nw); std::mt19937 rng(1); for(auto&x:h) x=rng();
  uint32_t *d,*C,*dl; CK(hipMalloc(&d,nw*4)); CK(hipMalloc(&C,(size_t)R*R*4)); CK(hipMalloc(&dl,(size_t)(R/128)*(R/128)*4));
  CK(hipMemcpy(d,h.data(),nw*4,hipMemcpyHostToDevice)); CK(hipMemset(C,0xff,(size_t)R*R*4));
  dim3 grid(R/128,R/128), block(twk::COUNT_THREADS);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bad_total=0;
  auto check=[&](const char* what,int diag){
    std::vector<uint32_t> hc((size_t)R*R); CK(hipMemcpy(hc.data(),C,(size_t)R*R*4,hipMemcpyDeviceToHost));
    int bad=0; std::mt19937 r2(7);
    for(int s=0;s<2000;++s){ uint32_t i=r2()%R,j=r2()%R; if(diag && j/128<i/128) continue; uint32_t ref=0; for(uint32_t k=0;k<W;++k) ref+=__builtin_popcount(h[(size_t)i*W+k]&h[(size_t)j*W+k]); if(ref!=hc[(size_t)i*R+j]){ if(bad<5) printf("MISMATCH %s (%u,%u) ref %u got %u\n",what,i,j,ref,hc[(size_t)i*R+j]); ++bad; } }
    printf("check %s: %d mismatches\n",what,bad); bad_total+=bad;
  };
  for(int diag=0; diag<2; ++diag){
    double tiles = diag? (double)(R/128)*(R/128+1)/2 : (double)(R/128)*(R/128);
    double wordops = tiles*128*128*W;
    for(int mode=0; mode<3; ++mode){     // 0 grid, 1 list row-major, 2 list patch order
      std::vector<uint32_t> list; twk::CountWork w{};
      if(mode){ list=make_list(R/128,diag,P,mode==2); CK(hipMemcpy(dl,list.data(),list.size()*4,hipMemcpyHostToDevice));
        w.rows=d; w.W=W; w.rowA0=0; w.rowB0=0; w.tiles=dl; w.n_tiles=(uint32_t)list.size(); w.n_rounds=(uint32_t)(list.size()/P); w.C=C; w.ldc=R; }
      auto launch=[&](){
        if(!mode){ hipLaunchKernelGGL((twk::k_count_tile_t<twk::COUNT_NW>),grid,block,0,0,d,W,0u,0u,diag,C,R); return; }
        const uint32_t first=w.n_rounds*P;
        if(first<w.n_tiles) hipLaunchKernelGGL(twk::k_zero_tiles,dim3(w.n_tiles-first),dim3(256),0,0,w.tiles,first,C,R);
        hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW>),dim3(P),block,0,0,w);
      };
      CK(hipMemset(C,0xff,(size_t)R*R*4));
      launch(); CK(hipDeviceSynchronize());
      check(mode==0?"grid":mode==1?"list":"list/patch",diag);
      float best=1e30f;
      for(int i=0;i<reps;++i){ CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
      printf("%-10s diag=%d R=%u W=%u P=%u tiles=%.0f best %.3f ms  word-pairs/s %.3e  VALU lane-ops/s %.3e (%.1f%% of 7.86e13; %.1f%% of the and+bcnt ceiling 2.62e13)\n",
             mode==0?"grid":mode==1?"list":"list/patch",diag,R,W,P,tiles,best,wordops/best*1e3,2*wordops/best*1e3,2*wordops/best*1e3/7.864e13*100,wordops/best*1e3/2.6214e13*100);
    }
  }
  return bad_total!=0;
}
