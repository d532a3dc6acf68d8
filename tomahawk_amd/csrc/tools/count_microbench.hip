// Stand-alone micro-benchmark + self-check of the count kernels (dev tool):
//   grid  k_count_tile_t   one block per 128 x 128 tile (2-D grid; the round-1 kernel)
//   list  k_count_list_t   persistent blocks over a tile list, data-parallel rounds + stream-K tail
// usage: count_microbench [rows=4096] [words=3136] [reps=3] [blocks=512]
//   ONLYMODE=2: only the list kernel in patch order on the rectangle (rocprofv3 counter passes)
//   NOSTORE=1: the list kernel without its epilogue (what the C stores cost: nothing measurable - 87.7 -> 88.3 % of the
//   ceiling at 5 chunks a tile; the self-check then fails by design)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../hip/ld_count.hip.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s @%d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)

static std::vector<uint32_t> make_list(uint32_t g, int diag, bool patch) {
	std::vector<uint32_t> seq;
	if (!patch) { for (uint32_t y = 0; y < g; ++y) for (uint32_t x = diag ? y : 0; x < g; ++x) seq.push_back(y << 16 | x); return seq; }
	for (uint32_t py = 0; py < g; py += 8) for (uint32_t px = 0; px < g; px += 8)
		for (uint32_t y = py; y < std::min(py + 8, g); ++y) for (uint32_t x = std::max(px, diag ? y : 0u); x < std::min(px + 8, g); ++x) seq.push_back(y << 16 | x);
	return seq;
}
int main(int argc,char**argv){
  uint32_t R = argc>1? atoi(argv[1]) : 4096;     // rows (multiple of 128)
  uint32_t W = argc>2? atoi(argv[2]) : 3136;     // words per row (multiple of 32)
  int reps = argc>3? atoi(argv[3]) : 3;
  uint32_t P = argc>4? atoi(argv[4]) : 512;      // persistent blocks
  size_t nw=(size_t)R*W;
  std::vector<uint32_t> h(nw); std::mt19937 rng(1); for(auto&x:h) x=rng();
  if(getenv("FUSED") && getenv("LIVE")) for(uint32_t r=0;r<R;++r) for(uint32_t k=atoi(getenv("LIVE"));k<W;++k) h[(size_t)r*W+k]=0;      // zero padding behind the live words
  if(getenv("FUSED")) for(uint32_t r=1;r<R;r+=2) for(uint32_t k=0;k<W;++k) h[(size_t)r*W+k] &= ~h[(size_t)(r-1)*W+k];   // rows 2v / 2v + 1 as a variant's H / Q planes: disjoint
  uint32_t *d,*C,*dl; twk::CountUnit* du; CK(hipMalloc(&d,nw*4)); CK(hipMalloc(&C,(size_t)R*R*4)); CK(hipMalloc(&dl,(size_t)(R/128)*(R/128)*4)); CK(hipMalloc(&du,(size_t)(R/128)*(R/128)*64*16+65536));
  const uint32_t min_chunks = getenv("MINCHUNKS")? atoi(getenv("MINCHUNKS")) : 8; uint32_t first_split=0; std::vector<twk::CountUnit> units;
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
  if(const char* gy=getenv("GRIDY")){   // probe: the 2-D grid kernel over gy tile rows only (e.g. 16 x 32 = 512 tiles = one round)
    const uint32_t ny=atoi(gy); dim3 g2(R/128,ny);
    hipLaunchKernelGGL((twk::k_count_tile_t<twk::COUNT_NW>),g2,block,0,0,d,W,0u,0u,0,C,R); CK(hipDeviceSynchronize());
    float best=1e30f;
    for(int i=0;i<reps;++i){ CK(hipEventRecord(e0)); hipLaunchKernelGGL((twk::k_count_tile_t<twk::COUNT_NW>),g2,block,0,0,d,W,0u,0u,0,C,R); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
    const double tiles=(double)(R/128)*ny, wordops=tiles*128*128*W;
    printf("grid %ux%u tiles=%.0f best %.3f ms  word-pairs/s %.3e (%.1f%% of the and+bcnt ceiling)\n",R/128,ny,tiles,best,wordops/best*1e3,wordops/best*1e3/2.6214e13*100);
    return 0;
  }
  uint32_t* tick; CK(hipMalloc(&tick,4));
  if(const char* bs=getenv("BASE_SWEEP")){   // probe: does the kernel's speed depend on where the rows lie?  The same rows at n different
    // offsets (multiples of STEP bytes, default 2 MiB - what separates one process's allocation from another's) inside one big
    // buffer; 3 warm launches, then the best of `reps`, per offset.
    const int n=atoi(bs); const size_t step=getenv("STEP")? strtoull(getenv("STEP"),nullptr,10) : (2ull<<20);
    uint8_t* big; CK(hipMalloc(&big,nw*4+(size_t)n*step));
    std::vector<uint32_t> list=make_list(R/128,0,true); CK(hipMemcpy(dl,list.data(),list.size()*4,hipMemcpyHostToDevice));
    twk::CountWork w{}; w.W=W; w.tiles=dl; w.C=C; w.ldc=R; w.ticket=tick;
    first_split=twk::build_count_units((uint32_t)list.size(),W/twk::KC,P,min_chunks,units,8,8); twk::fill_unit_tiles(units,list.data());
    CK(hipMemcpy(du,units.data(),units.size()*16,hipMemcpyHostToDevice)); w.units=du; w.n_units=(uint32_t)units.size(); w.n_queues=1; w.queue_begin[0]=0; w.queue_begin[1]=w.n_units;
    std::vector<float> ms_all;
    for(int o=0;o<n;++o){
      uint32_t* base=reinterpret_cast<uint32_t*>(big+(size_t)o*step);
      CK(hipMemcpy(base,d,nw*4,hipMemcpyDeviceToDevice)); w.rows=base;
      float best=1e30f;
      for(int i=0;i<3+reps;++i){ CK(hipMemsetAsync(tick,0,4,0)); if(first_split<list.size()) hipLaunchKernelGGL(twk::k_zero_tiles,dim3((uint32_t)list.size()-first_split),dim3(256),0,0,w.tiles,first_split,C,R);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW>),dim3(P),block,0,0,w); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(i>=3&&ms<best)best=ms; }
      ms_all.push_back(best);
    }
    float mn=1e30f,mx=0; for(float x:ms_all){ mn=std::min(mn,x); mx=std::max(mx,x); }
    printf("R=%u W=%u, rows at %d offsets of %zu bytes: best launch %.3f .. %.3f ms (ratio %.2f)\n  ",R,W,n,step,mn,mx,mx/mn);
    for(int o=0;o<n;++o) printf("%.2f%s",ms_all[o],(o%16==15)?"\n  ":" "); printf("\n");
    return 0;
  }
  if(getenv("FINISH")){   // probe: per-block finish times of the static list kernel (2 full rounds), by XCD
    std::vector<uint32_t> list=make_list(R/128,0,true); CK(hipMemcpy(dl,list.data(),list.size()*4,hipMemcpyHostToDevice));
    twk::CountWork w{}; w.rows=d; w.W=W; w.tiles=dl; w.C=C; w.ldc=R; w.ticket=tick; first_split=twk::build_count_units((uint32_t)list.size(),W/twk::KC,P,min_chunks,units,getenv("SHAREDIV")?atoi(getenv("SHAREDIV")):8,getenv("TAILROUNDS")?atoi(getenv("TAILROUNDS")):8); twk::fill_unit_tiles(units,list.data()); CK(hipMemcpy(du,units.data(),units.size()*16,hipMemcpyHostToDevice)); w.units=du; w.n_units=(uint32_t)units.size(); w.n_queues=1; w.queue_begin[0]=0; w.queue_begin[1]=w.n_units;
    const int probe_reps = getenv("FINISH_REPS")? atoi(getenv("FINISH_REPS")) : 2;       // launches back to back; the last one is reported
    for(int rep=0;rep<probe_reps;++rep){ CK(hipMemset(tick,0,4)); if(first_split<list.size()) hipLaunchKernelGGL(twk::k_zero_tiles,dim3((uint32_t)list.size()-first_split),dim3(256),0,0,w.tiles,first_split,C,R); hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW,5>),dim3(P),block,0,0,w); }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> o4(4*P); CK(hipMemcpy(o4.data(),C,o4.size()*8,hipMemcpyDeviceToHost));
    std::vector<unsigned long long> o(2*P); for(uint32_t b=0;b<P;++b){ o[2*b]=o4[4*b]; o[2*b+1]=o4[4*b+1]; }
    { // the shader clock the blocks ran at: s_memtime ticks per tick of the constant 100 MHz counter, over each block's life
      double mn=1e30,mx=0,sum=0,life=0; for(uint32_t b=0;b<P;++b){ const double f=(double)o4[4*b+3]/(double)o4[4*b+2]*100.0; mn=std::min(mn,f); mx=std::max(mx,f); sum+=f; life+=o4[4*b+2]/100.0; }
      printf("shader clock over the blocks' lives (R=%u W=%u, second of two launches): mean %.0f MHz (min %.0f, max %.0f); mean block life %.1f us\n",R,W,sum/P,mn,mx,life/P); }
    unsigned long long t0=~0ull,t1=0; for(uint32_t b=0;b<P;++b){ t0=std::min(t0,o[2*b]); t1=std::max(t1,o[2*b]); }
    printf("finish-time spread over %u blocks: %.1f us (wall_clock64 ticks at 100 MHz)\n",P,(t1-t0)/100.0);
    for(int x=0;x<8;++x){ double mn=1e30,mx=0,sum=0; int n=0; for(uint32_t b=0;b<P;++b) if((int)(o[2*b+1]>>32&15)==x){ double t=(o[2*b]-t0)/100.0; mn=std::min(mn,t); mx=std::max(mx,t); sum+=t; ++n; }
      printf("  XCD %d: %3d blocks, finish offset min %.1f mean %.1f max %.1f us\n",x,n,mn,n?sum/n:0,mx); }
    // same-CU pairs
    printf("  first 16 blocks: "); for(uint32_t b=0;b<16;++b) printf("[b%u xcc%llu hw%05llx t%.0f] ",b,o[2*b+1]>>32&15,o[2*b+1]&0xFFFFF,(o[2*b]-t0)/100.0); printf("\n");
    return 0;
  }
  if(getenv("FUSED")){   // the fused count -> screen kernels on whole tiles (short rows), with their epilogue and without it (EXPERIMENT 6):
    // what does the screen cost next to the K loop?  Random rows, so nothing passes the screen.  LIVE=<words>: the row's live words (last chunk cut short)
    std::vector<uint32_t> list=make_list(R/128,0,true); CK(hipMemcpy(dl,list.data(),list.size()*4,hipMemcpyHostToDevice));
    const uint32_t live = getenv("LIVE")? atoi(getenv("LIVE")) : W;
    twk::CountWork w{}; w.rows=d; w.W=W; w.tiles=dl; w.C=C; w.ldc=R; w.ticket=tick;
    { const uint32_t live_last = live - (W/twk::KC - 1)*twk::KC; w.last_halves = (live<W && W-live<twk::KC) ? (live_last+1)/2 : 0; if(w.last_halves>12) w.last_halves=0; }
    twk::build_count_units((uint32_t)list.size(),W/twk::KC,P,W/twk::KC+1,units,8,8); twk::fill_unit_tiles(units,list.data());
    CK(hipMemcpy(du,units.data(),units.size()*16,hipMemcpyHostToDevice)); w.units=du; w.n_units=(uint32_t)units.size(); w.n_queues=1; w.queue_begin[0]=0; w.queue_begin[1]=w.n_units;
    std::vector<uint32_t> pop(R+256); for(uint32_t r=0;r<R;++r){ uint32_t c=0; for(uint32_t k=0;k<W;++k) c+=__builtin_popcount(h[(size_t)r*W+k]); pop[r]=c; }
    uint32_t* dpop; CK(hipMalloc(&dpop,pop.size()*4)); CK(hipMemcpy(dpop,pop.data(),pop.size()*4,hipMemcpyHostToDevice));
    unsigned long long* dn; CK(hipMalloc(&dn,64)); CK(hipMemset(dn,0,64));
    twk::ScreenWork sw{}; sw.rowpop=dpop; sw.a0=0; sw.b0=0; sw.n_variants=R; sw.diag=0; sw.col_hi=nullptr; sw.list_zone=0; sw.probe_zone=0; sw.cut=0.1*(1.0-1e-6);
    sw.cand=C; sw.cap=(unsigned long long)R*R/8; sw.n_cand=dn; sw.chunk=64;
    twk::ScreenWork* dsw; CK(hipMalloc(&dsw,sizeof(sw)));
    float4* dterms; CK(hipMalloc(&dterms,(size_t)(R+256)*sizeof(float4)));
    const double tiles=(double)list.size();
    for(int form=0; form<3; ++form){   // 0 phased (rows = variants), 1 unphased four products, 2 unphased three products (rows 2v, 2v+1 = H, Q)
      sw.nA=sw.nB= form? R/2 : R; sw.n_variants=sw.nA; sw.two_n = form? 2.0*W*32 : 1.0*W*32;
      hipLaunchKernelGGL(twk::k_screen_terms,dim3((sw.nA+255)/256),dim3(256),0,0,(const uint32_t*)dpop,sw.nA,form?2:1,sw.two_n,sw.cut,dterms);
      sw.terms=dterms; sw.slack=0.5f+(float)sw.two_n*(1.0f/1048576.0f);
      CK(hipMemcpy(dsw,&sw,sizeof(sw),hipMemcpyHostToDevice));
      for(int noepi=0; noepi<2; ++noepi){
        auto launch=[&](){ CK(hipMemsetAsync(tick,0,4,0)); CK(hipMemsetAsync(dn,0,8,0));
          if(form==0){ if(noepi) hipLaunchKernelGGL((twk::k_count_screen_t<twk::COUNT_NW,6>),dim3(P),block,0,0,w,dsw); else hipLaunchKernelGGL((twk::k_count_screen_t<twk::COUNT_NW,0>),dim3(P),block,0,0,w,dsw); }
          else if(form==1){ if(noepi) hipLaunchKernelGGL((twk::k_count_screen_unphased_t<twk::COUNT_NW,6>),dim3(P),block,0,0,w,dsw); else hipLaunchKernelGGL((twk::k_count_screen_unphased_t<twk::COUNT_NW,0>),dim3(P),block,0,0,w,dsw); }
          else { if(noepi) hipLaunchKernelGGL((twk::k_count3_screen_unphased_t<twk::COUNT_NW,6>),dim3(P),block,0,0,w,dsw); else hipLaunchKernelGGL((twk::k_count3_screen_unphased_t<twk::COUNT_NW,0>),dim3(P),block,0,0,w,dsw); } };
        launch(); launch(); CK(hipDeviceSynchronize());
        float best=1e30f;
        for(int i=0;i<reps;++i){ CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
        unsigned long long nc=0; CK(hipMemcpy(&nc,dn,8,hipMemcpyDeviceToHost));
        const uint32_t words_done = w.last_halves? (W/twk::KC-1)*twk::KC + 2*w.last_halves : W;
        const double prod = tiles*128*128*words_done*(form==2?0.75:1.0), equiv = prod;
        printf("fused %-22s %-12s R=%u W=%u live=%u tiles=%.0f best %.3f ms  products/s %.3e  (%.1f%% of the and+bcnt ceiling 2.62e13)  candidates %llu\n",
               form==0?"phased":form==1?"unphased four-product":"unphased three-product", noepi?"no epilogue":"with screen", R,W,live,tiles,best,prod/best*1e3,equiv/best*1e3/2.6214e13*100,nc);
      }
    }
    return 0;
  }
  if(getenv("THREE")){   // the three-product form of the unphased planes (rows 2v / 2v + 1 = H / Q of variant v): k_count3_list_t against k_count_list_t on
    // the same rows, patch order, rectangle; checked against the host on sampled variant pairs
    std::vector<uint32_t> list=make_list(R/128,0,true); CK(hipMemcpy(dl,list.data(),list.size()*4,hipMemcpyHostToDevice));
    twk::CountWork w{}; w.rows=d; w.W=W; w.tiles=dl; w.C=C; w.ldc=R; w.ticket=tick;
    first_split=twk::build_count_units((uint32_t)list.size(),W/twk::KC,P,min_chunks,units,8,8); twk::fill_unit_tiles(units,list.data());
    CK(hipMemcpy(du,units.data(),units.size()*16,hipMemcpyHostToDevice)); w.units=du; w.n_units=(uint32_t)units.size(); w.n_queues=1; w.queue_begin[0]=0; w.queue_begin[1]=w.n_units;
    const double tiles=(double)list.size(), wordops=tiles*128*128*W;
    if(const char* ex=getenv("EXPER")){   // timing only (wrong counts): the three- and four-product list kernels without the chunk barrier (7) / without the operand staging (8)
      const int which=atoi(ex);
      for(int three=0; three<2; ++three){
        auto launch=[&](){ CK(hipMemsetAsync(tick,0,4,0));
          if(which==7){ if(three) hipLaunchKernelGGL((twk::k_count3_list_t<twk::COUNT_NW,7>),dim3(P),block,0,0,w); else hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW,7>),dim3(P),block,0,0,w); }
          else { if(three) hipLaunchKernelGGL((twk::k_count3_list_t<twk::COUNT_NW,8>),dim3(P),block,0,0,w); else hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW,8>),dim3(P),block,0,0,w); } };
        launch(); CK(hipDeviceSynchronize()); float best=1e30f;
        for(int i=0;i<reps;++i){ CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
        const double exec=three? wordops*0.75 : wordops;
        printf("EXPERIMENT %d %-13s best %.3f ms  executed word-pairs/s %.3e (%.1f%% of the and+bcnt ceiling)\n",which,three?"three-product":"four-product",best,exec/best*1e3,exec/best*1e3/2.6214e13*100);
      }
      return 0;
    }
    for(int three=0; three<2; ++three){
      auto launch=[&](){ CK(hipMemsetAsync(tick,0,4,0));
        const uint32_t fs = first_split;
        if(fs<list.size()) hipLaunchKernelGGL(twk::k_zero_tiles,dim3((uint32_t)list.size()-fs),dim3(256),0,0,w.tiles,fs,C,R,three?64u:128u);
        if(three) hipLaunchKernelGGL((twk::k_count3_list_t<twk::COUNT_NW>),dim3(P),block,0,0,w); else hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW>),dim3(P),block,0,0,w); };
      CK(hipMemset(C,0xff,(size_t)R*R*4)); launch(); CK(hipDeviceSynchronize());
      if(three){ std::vector<uint32_t> hc((size_t)R*R/2); CK(hipMemcpy(hc.data(),C,hc.size()*4,hipMemcpyDeviceToHost)); int bad=0; std::mt19937 r2(7);
        for(int s=0;s<2000;++s){ const uint32_t a=r2()%(R/2), b=r2()%(R/2); uint32_t hh=0, ss=0;
          for(uint32_t k=0;k<W;++k){ const uint32_t ha=h[(size_t)(2*a)*W+k], qa=h[(size_t)(2*a+1)*W+k], hb=h[(size_t)(2*b)*W+k], qb=h[(size_t)(2*b+1)*W+k];
            hh+=__builtin_popcount(ha&hb); ss+=__builtin_popcount(qa&(hb|qb))+__builtin_popcount((ha|qa)&qb); }
          const uint32_t g0=hc[(size_t)a*R+2*b], g1=hc[(size_t)a*R+2*b+1];
          if(g0!=hh||g1!=ss){ if(bad<5) printf("MISMATCH three (%u,%u) ref %u %u got %u %u\n",a,b,hh,ss,g0,g1); ++bad; } }
        printf("check %s: %d mismatches\n","three-product",bad); bad_total+=bad; }
      else check("four-product",0);
      float best=1e30f;
      for(int i=0;i<reps;++i){ CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
      const double exec=three? wordops*0.75 : wordops;
      printf("%-13s R=%u W=%u P=%u tiles=%.0f whole=%u units=%u best %.3f ms  variant pairs/s %.4e  executed word-pairs/s %.3e (%.1f%% of the and+bcnt ceiling 2.62e13)\n",
             three?"three-product":"four-product",R,W,P,tiles,first_split,w.n_units,best,tiles*64*64/best*1e3,exec/best*1e3,exec/best*1e3/2.6214e13*100);
    }
    return bad_total!=0;
  }
  const int only_mode = getenv("ONLYMODE")? atoi(getenv("ONLYMODE")) : -1;     // counter passes: one kernel form (2 = list/patch), rectangle only
  for(int diag=0; diag<(only_mode>=0?1:2); ++diag){
    double tiles = diag? (double)(R/128)*(R/128+1)/2 : (double)(R/128)*(R/128);
    double wordops = tiles*128*128*W;
    for(int mode=0; mode<3; ++mode){     // 0 grid, 1 list row-major, 2 list patch order, 3.. timing probes
      if(only_mode>=0 && mode!=only_mode) continue;
      std::vector<uint32_t> list; twk::CountWork w{};
      if(mode){ list=make_list(R/128,diag,mode>=2); CK(hipMemcpy(dl,list.data(),list.size()*4,hipMemcpyHostToDevice));
        w.rows=d; w.W=W; w.rowA0=0; w.rowB0=0; w.tiles=dl; w.C=C; w.ldc=R; w.ticket=tick; first_split=twk::build_count_units((uint32_t)list.size(),W/twk::KC,P,min_chunks,units,getenv("SHAREDIV")?atoi(getenv("SHAREDIV")):8,getenv("TAILROUNDS")?atoi(getenv("TAILROUNDS")):8); twk::fill_unit_tiles(units,list.data()); CK(hipMemcpy(du,units.data(),units.size()*16,hipMemcpyHostToDevice)); w.units=du; w.n_units=(uint32_t)units.size(); w.n_queues=1; w.queue_begin[0]=0; w.queue_begin[1]=w.n_units; }
      auto launch=[&](){
        if(!mode){ hipLaunchKernelGGL((twk::k_count_tile_t<twk::COUNT_NW>),grid,block,0,0,d,W,0u,0u,diag,C,R); return; }
        CK(hipMemsetAsync(tick,0,4,0));
        if(first_split<list.size()) hipLaunchKernelGGL(twk::k_zero_tiles,dim3((uint32_t)list.size()-first_split),dim3(256),0,0,w.tiles,first_split,C,R);
        if(getenv("NOSTORE")) hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW,6>),dim3(P),block,0,0,w);
        else hipLaunchKernelGGL((twk::k_count_list_t<twk::COUNT_NW>),dim3(P),block,0,0,w);
      };
      static const char* names[]={"grid","list","list/patch"};
      CK(hipMemset(C,0xff,(size_t)R*R*4));
      launch(); CK(hipDeviceSynchronize());
      check(names[mode],diag);
      float best=1e30f;
      for(int i=0;i<reps;++i){ CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
      printf("%-10s diag=%d R=%u W=%u P=%u tiles=%.0f whole=%u units=%u best %.3f ms  word-pairs/s %.3e  VALU lane-ops/s %.3e (%.1f%% of 7.86e13; %.1f%% of the and+bcnt ceiling 2.62e13)\n",
             names[mode],diag,R,W,P,tiles,mode?first_split:0u,mode?w.n_units:0u,best,wordops/best*1e3,2*wordops/best*1e3,2*wordops/best*1e3/7.864e13*100,wordops/best*1e3/2.6214e13*100);
    }
  }
  return bad_total!=0;
}
