// Dev tool (round 6): can gfx950's three-input boolean instruction take the v_or out of the three-product contraction?
//   S = popc(Q_A & (H_B | Q_B)) + popc((H_A | Q_A) & Q_B)      - each term is ONE v_bitop3_b32 (truth table 0xE0: a & (b | c)) + v_bcnt
// (1) the truth table's convention, checked on random words against a & (b | c);
// (2) the issue rate of (v_bitop3_b32, s_nop, v_bcnt-accumulate) against (v_and_b32, s_nop, v_bcnt-accumulate), register-only streams of 24
//     products per step, 4 waves per SIMD (512-thread blocks, two per CU, like the count kernel); and the kernel's own mix: 8 v_and + 16 v_bitop3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
#define P(T,A,B,C) "v_and_b32 v" #T ", v" #A ", v" #B "\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define B3(T,A,B,C2,C) "v_bitop3_b32 v" #T ", v" #A ", v" #B ", v" #C2 " bitop3:0xe0\n\ts_nop 0\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
// one B variant against four A variants: A words H v33 v35 v37 v39, Q v34 v36 v38 v40; B words H v41, Q v42; HH v64..67, S v68..71
#define AND12 P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) P(56,34,41,68) P(57,36,41,69) P(56,38,41,70) P(57,40,41,71) P(56,33,42,68) P(57,35,42,69) P(56,37,42,70) P(57,39,42,71)
#define BIT12 B3(56,33,41,42,64) B3(57,35,41,42,65) B3(56,37,41,42,66) B3(57,39,41,42,67) B3(56,34,41,42,68) B3(57,36,41,42,69) B3(56,38,41,42,70) B3(57,40,41,42,71) B3(56,42,33,34,68) B3(57,42,35,36,69) B3(56,42,37,38,70) B3(57,42,39,40,71)
#define MIX12 P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) B3(56,34,41,42,68) B3(57,36,41,42,69) B3(56,38,41,42,70) B3(57,40,41,42,71) B3(56,42,33,34,68) B3(57,42,35,36,69) B3(56,42,37,38,70) B3(57,42,39,40,71)
// ... the same streams with the kernel's LDS traffic beside them: twelve ds_read_b64 per 48 products (into registers the products do not read)
#define LDS12 "ds_read_b64 v[44:45], v72\n\tds_read_b64 v[46:47], v72 offset:128\n\tds_read_b64 v[48:49], v72 offset:2048\n\tds_read_b64 v[50:51], v72 offset:2176\n\t" \
              "ds_read_b64 v[52:53], v72 offset:4096\n\tds_read_b64 v[54:55], v72 offset:4224\n\tds_read_b64 v[58:59], v72 offset:6144\n\tds_read_b64 v[60:61], v72 offset:6272\n\t" \
              "ds_read_b64 v[44:45], v73\n\tds_read_b64 v[46:47], v73 offset:128\n\tds_read_b64 v[48:49], v73 offset:2048\n\tds_read_b64 v[50:51], v73 offset:2176\n\ts_waitcnt lgkmcnt(12)\n\t"
#define LDS8 "ds_read_b64 v[44:45], v72\n\tds_read_b64 v[46:47], v72 offset:128\n\tds_read_b64 v[48:49], v72 offset:2048\n\tds_read_b64 v[50:51], v72 offset:2176\n\t" \
             "ds_read_b64 v[52:53], v73 offset:4096\n\tds_read_b64 v[54:55], v73 offset:4224\n\tds_read_b64 v[58:59], v73 offset:6144\n\tds_read_b64 v[60:61], v73 offset:6272\n\ts_waitcnt lgkmcnt(8)\n\t"
#define LDS4 "ds_read_b64 v[44:45], v72\n\tds_read_b64 v[46:47], v72 offset:128\n\tds_read_b64 v[52:53], v73 offset:4096\n\tds_read_b64 v[54:55], v73 offset:4224\n\ts_waitcnt lgkmcnt(4)\n\t"
#define LDS6x128 "ds_read_b128 v[44:47], v72\n\tds_read_b128 v[48:51], v72 offset:2048\n\tds_read_b128 v[52:55], v72 offset:4096\n\tds_read_b128 v[58:61], v72 offset:6144\n\t" \
             "ds_read_b128 v[44:47], v73\n\tds_read_b128 v[48:51], v73 offset:2048\n\ts_waitcnt lgkmcnt(6)\n\t"
#define R8(X) X X X X X X X X
template<int MODE> __global__ __launch_bounds__(512, 2) void k(uint32_t* out, int iters){
  __shared__ uint32_t lds[16384];
  if(MODE>=3){ for(int i=threadIdx.x;i<16384;i+=512) lds[i]=i; __syncthreads();
    const uint32_t a=(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x>>3&7)*256 + ((threadIdx.x>>3&7)<<4), b=a+8192+((threadIdx.x&7)*256);
    asm volatile("v_mov_b32 v72, %0\n\tv_mov_b32 v73, %1" :: "v"(a), "v"(b) : "v72", "v73"); }
  for(int it=0; it<iters; ++it){
    if(MODE==0) asm volatile(R8(AND12 AND12) ::: CLOB);
    else if(MODE==1) asm volatile(R8(BIT12 BIT12) ::: CLOB);
    else if(MODE==2) asm volatile(R8(MIX12 MIX12) ::: CLOB);
    else if(MODE==3) asm volatile(R8(LDS12 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
    else if(MODE==4) asm volatile(R8(LDS12 MIX12 MIX12 MIX12 MIX12) ::: CLOB, "v72", "v73");
    else if(MODE==5) asm volatile(R8(LDS8 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
    else if(MODE==6) asm volatile(R8(LDS4 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
    else asm volatile(R8(LDS6x128 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
  }
  uint32_t s; asm volatile("v_add_u32 %0, v64, v68" : "=v"(s)); out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE> void run(const char* name){
  int blocks=256*2; uint32_t* d; CK(hipMalloc(&d,(size_t)blocks*512*4));
  int iters=20000; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,d,2000); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,d,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  const double steps=(double)blocks*512*iters*8*(MODE>=3?2:1);
  printf("%-58s %.3f ms  products/s %.3e (%.1f %% of the and+bcnt ceiling 2.62e13)\n",name,ms,steps*24/ms*1e3,steps*24/ms*1e3/2.6214e13*100); fflush(stdout);
}
__global__ void k_check(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* o, int n){
  const int i=blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  uint32_t r; asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe0" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
  o[i]=r;
}
int main(){
  const int n=1<<20; std::vector<uint32_t> a(n),b(n),c(n),o(n); std::mt19937 rng(3); for(int i=0;i<n;++i){a[i]=rng();b[i]=rng();c[i]=rng();}
  uint32_t *da,*db,*dc,*dout; CK(hipMalloc(&da,n*4)); CK(hipMalloc(&db,n*4)); CK(hipMalloc(&dc,n*4)); CK(hipMalloc(&dout,n*4));
  CK(hipMemcpy(da,a.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(db,b.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dc,c.data(),n*4,hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check,dim3(n/256),dim3(256),0,0,da,db,dc,dout,n); CK(hipMemcpy(o.data(),dout,n*4,hipMemcpyDeviceToHost));
  int bad=0; for(int i=0;i<n;++i) if(o[i]!=(a[i]&(b[i]|c[i]))) ++bad;
  printf("v_bitop3_b32 bitop3:0xe0 == a & (b | c): %d mismatches of %d words\n",bad,n);
  run<0>("24 x (v_and_b32, s_nop, v_bcnt)");
  run<1>("24 x (v_bitop3_b32, s_nop, v_bcnt)");
  run<2>("8 x v_and + 16 x v_bitop3 (the three-product mix, no v_or)");
  run<3>("48 x v_and products + 12 ds_read_b64");
  run<4>("48 products of the mix + 12 ds_read_b64");
  run<5>("48 x v_and products + 8 ds_read_b64");
  run<6>("48 x v_and products + 4 ds_read_b64");
  run<7>("48 x v_and products + 6 ds_read_b128 (the bytes of 12 b64)");
  return bad!=0;
}
