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
// ... and the same twelve reads spread through the products instead of in front of them: three in front of every AND12 (LDS3a..d), one after every fourth
// product (PL: a product with a read after it), and one IN THE PLACE of a product's s_nop (PN)
#define RD(D0,D1,A,OFF) "ds_read_b64 v[" #D0 ":" #D1 "], v" #A " offset:" #OFF "\n\t"
#define LDS3a RD(44,45,72,0) RD(46,47,72,128) RD(48,49,72,2048)
#define LDS3b RD(50,51,72,2176) RD(52,53,72,4096) RD(54,55,72,4224)
#define LDS3c RD(58,59,72,6144) RD(60,61,72,6272) RD(44,45,73,0)
#define LDS3d RD(46,47,73,128) RD(48,49,73,2048) RD(50,51,73,2176) "s_waitcnt lgkmcnt(12)\n\t"
#define PN(T,A,B,C,D0,D1,AD,OFF) "v_and_b32 v" #T ", v" #A ", v" #B "\n\tds_read_b64 v[" #D0 ":" #D1 "], v" #AD " offset:" #OFF "\n\tv_bcnt_u32_b32 v" #C ", v" #T ", v" #C "\n\t"
#define PL(T,A,B,C,D0,D1,AD,OFF) P(T,A,B,C) RD(D0,D1,AD,OFF)
#define AND12_L3(R0,R1,R2) P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) R0 P(57,39,41,67) P(56,34,41,68) P(57,36,41,69) P(56,38,41,70) R1 P(57,40,41,71) P(56,33,42,68) P(57,35,42,69) P(56,37,42,70) R2 P(57,39,42,71)
#define AND12_N3(D0,D1,D2,D3,D4,D5,AD,O0,O1,O2) PN(56,33,41,64,D0,D1,AD,O0) P(57,35,41,65) P(56,37,41,66) P(57,39,41,67) PN(56,34,41,68,D2,D3,AD,O1) P(57,36,41,69) P(56,38,41,70) P(57,40,41,71) PN(56,33,42,68,D4,D5,AD,O2) P(57,35,42,69) P(56,37,42,70) P(57,39,42,71)
#define MIX12_L3(R0,R1,R2) P(56,33,41,64) P(57,35,41,65) P(56,37,41,66) R0 P(57,39,41,67) B3(56,34,41,42,68) B3(57,36,41,42,69) B3(56,38,41,42,70) R1 B3(57,40,41,42,71) B3(56,42,33,34,68) B3(57,42,35,36,69) B3(56,42,37,38,70) R2 B3(57,42,39,40,71)
#define R8(X) X X X X X X X X
// (3) the half-slot as the count kernel runs it (bitop3_probe_gen.h): the reads of the next half-slot into one register set, the 48 products of this one
//     from the other - with v_and in the place of every v_bitop3 (timing only) and as it is; and the same with the reads landing apart from the products' sources
#include "bitop3_probe_gen.h"
#define CLOB3 "v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
#define CLOB2 "v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
template<int MODE> __global__ __launch_bounds__(512, 2) void k(uint32_t* out, int iters){
  __shared__ uint32_t lds[16384];
  if(MODE>=3){ for(int i=threadIdx.x;i<16384;i+=512) lds[i]=i; __syncthreads();
    const uint32_t a=(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x>>3&7)*256 + ((threadIdx.x>>3&7)<<4), b=(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + 8192 + (threadIdx.x&7)*256 + ((threadIdx.x&7)<<4);      // (the count kernel's PAIRED lane rows: no bank conflicts)
    asm volatile("v_mov_b32 v72, %0\n\tv_mov_b32 v73, %1\n\tv_add_u32 v74, 2048, v73\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %1" :: "v"(a), "v"(b) : "v72", "v73", "v74", "v10", "v11"); }
  for(int it=0; it<iters; ++it){
    if(MODE==0) asm volatile(R8(AND12 AND12) ::: CLOB);
    else if(MODE==1) asm volatile(R8(BIT12 BIT12) ::: CLOB);
    else if(MODE==2) asm volatile(R8(MIX12 MIX12) ::: CLOB);
    else if(MODE==3) asm volatile(R8(LDS12 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
    else if(MODE==4) asm volatile(R8(LDS12 MIX12 MIX12 MIX12 MIX12) ::: CLOB, "v72", "v73");
    else if(MODE==5) asm volatile(R8(LDS8 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
    else if(MODE==6) asm volatile(R8(LDS4 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
    else if(MODE==7) asm volatile(R8(LDS6x128 AND12 AND12 AND12 AND12) ::: CLOB, "v72", "v73");
    else if(MODE==8) asm volatile(R8(LDS3a AND12 LDS3b AND12 LDS3c AND12 LDS3d AND12) ::: CLOB, "v72", "v73");
    else if(MODE==9) asm volatile(R8(AND12_L3(RD(44,45,72,0),RD(46,47,72,128),RD(48,49,72,2048)) AND12_L3(RD(50,51,72,2176),RD(52,53,72,4096),RD(54,55,72,4224))
                                     AND12_L3(RD(58,59,72,6144),RD(60,61,72,6272),RD(44,45,73,0)) AND12_L3(RD(46,47,73,128),RD(48,49,73,2048),RD(50,51,73,2176)) "s_waitcnt lgkmcnt(12)\n\t") ::: CLOB, "v72", "v73");
    else if(MODE==10) asm volatile(R8(AND12_N3(44,45,46,47,48,49,72,0,128,2048) AND12_N3(50,51,52,53,54,55,72,2176,4096,4224)
                                      AND12_N3(58,59,60,61,44,45,73,6144,6272,0) AND12_N3(46,47,48,49,50,51,73,128,2048,2176) "s_waitcnt lgkmcnt(12)\n\t") ::: CLOB, "v72", "v73");
    else if(MODE==12) asm volatile(STEP_AND STEP_AND STEP_AND STEP_AND ::: CLOB, CLOB2, "v72", "v73");
    else if(MODE==13) asm volatile(STEP_BITOP3 STEP_BITOP3 STEP_BITOP3 STEP_BITOP3 ::: CLOB, CLOB2, "v72", "v73");
    else if(MODE==14) asm volatile(STEP_AND_APART STEP_AND_APART STEP_AND_APART STEP_AND_APART ::: CLOB, CLOB2, "v72", "v73");
    else if(MODE==15) asm volatile(STEP_BITOP3_APART STEP_BITOP3_APART STEP_BITOP3_APART STEP_BITOP3_APART ::: CLOB, CLOB2, "v72", "v73");
    else if(MODE==16) asm volatile(STEP_AND_APART_SWAPB STEP_AND_APART_SWAPB STEP_AND_APART_SWAPB STEP_AND_APART_SWAPB ::: CLOB, CLOB2, "v72", "v73");
    else if(MODE==17) asm volatile(STEP_BITOP3_APART_SWAPB STEP_BITOP3_APART_SWAPB STEP_BITOP3_APART_SWAPB STEP_BITOP3_APART_SWAPB ::: CLOB, CLOB2, "v72", "v73");
    else if(MODE==18) asm volatile(STEP_AND_SWAPB STEP_AND_SWAPB STEP_AND_SWAPB STEP_AND_SWAPB ::: CLOB, CLOB2, "v72", "v73", "v74");
    else if(MODE==19) asm volatile(STEP_BITOP3_SWAPB STEP_BITOP3_SWAPB STEP_BITOP3_SWAPB STEP_BITOP3_SWAPB ::: CLOB, CLOB2, "v72", "v73", "v74");
    else if(MODE==20) asm volatile(STEP_BITOP3_SWAPB_TUPLEA STEP_BITOP3_SWAPB_TUPLEA STEP_BITOP3_SWAPB_TUPLEA STEP_BITOP3_SWAPB_TUPLEA ::: CLOB, CLOB2, "v72", "v73", "v74", "v75");
    else if(MODE==21) asm volatile(STEP_BITOP3_TUPLEA STEP_BITOP3_TUPLEA STEP_BITOP3_TUPLEA STEP_BITOP3_TUPLEA ::: CLOB, CLOB2, "v72", "v73", "v74", "v75");
    else if(MODE==22) asm volatile(STEP_AND_WIDEA STEP_AND_WIDEA STEP_AND_WIDEA STEP_AND_WIDEA ::: CLOB3, CLOB, CLOB2, "v72", "v73", "v74");
    else if(MODE==23) asm volatile(STEP_BITOP3_WIDEA STEP_BITOP3_WIDEA STEP_BITOP3_WIDEA STEP_BITOP3_WIDEA ::: CLOB3, CLOB, CLOB2, "v72", "v73", "v74");
    else if(MODE==24) asm volatile(STEP_AND_SLOTS STEP_AND_SLOTS STEP_AND_SLOTS STEP_AND_SLOTS ::: CLOB3, CLOB, CLOB2, "v72", "v73", "v74", "v75", "v10", "v11");
    else if(MODE==25) asm volatile(STEP_BITOP3_SLOTS STEP_BITOP3_SLOTS STEP_BITOP3_SLOTS STEP_BITOP3_SLOTS ::: CLOB3, CLOB, CLOB2, "v72", "v73", "v74", "v75", "v10", "v11");
    else asm volatile(R8(MIX12_L3(RD(44,45,72,0),RD(46,47,72,128),RD(48,49,72,2048)) MIX12_L3(RD(50,51,72,2176),RD(52,53,72,4096),RD(54,55,72,4224))
                                     MIX12_L3(RD(58,59,72,6144),RD(60,61,72,6272),RD(44,45,73,0)) MIX12_L3(RD(46,47,73,128),RD(48,49,73,2048),RD(50,51,73,2176)) "s_waitcnt lgkmcnt(12)\n\t") ::: CLOB, "v72", "v73");
  }
  uint32_t s; asm volatile("v_add_u32 %0, v64, v68\n\tv_add_u32 %0, %0, v12\n\tv_add_u32 %0, %0, v20" : "=v"(s)); out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE> void run(const char* name){
  int blocks=256*2; uint32_t* d; CK(hipMalloc(&d,(size_t)blocks*512*4));
  int iters=20000; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,d,2000); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>),dim3(blocks),dim3(512),0,0,d,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  const double steps=(double)blocks*512*iters*(MODE>=22 ? 32 : MODE>=12 ? 16 : 8*(MODE>=3?2:1));      // (in units of 24 products)
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
  run<8>("48 x v_and products, 3 ds_read_b64 in front of every 12");
  run<9>("48 x v_and products, a ds_read_b64 after every 4th");
  run<10>("48 x v_and products, a ds_read_b64 for every 4th s_nop");
  run<11>("48 products of the mix, a ds_read_b64 after every 4th");
  run<12>("half-slots as in the kernel, v_and for every v_bitop3");
  run<13>("half-slots as in the kernel (8 v_and + 16 v_bitop3 per 24)");
  run<14>("... v_and only, reads landing apart from the sources");
  run<15>("... the mix, reads landing apart from the sources");
  run<16>("... v_and only, reads apart, hA qA hB qB in four banks");
  run<17>("... the mix, reads apart, hA qA hB qB in four banks");
  run<18>("half-slots, v_and only, B pairs by swapped ds_read2_b32");
  run<19>("half-slots, the mix, B pairs by swapped ds_read2_b32");
  run<20>("... and the A variants' H Q by one ds_read2_b64 each");
  run<21>("half-slots, the mix, ds_read2_b64 for A, B plain (two in a bank)");
  run<22>("whole A slots by ds_read_b128, swapped B pairs, v_and only");
  run<23>("whole A slots by ds_read_b128, swapped B pairs, the mix");
  run<24>("whole A and B slots by ds_read_b128 (B image swapped), v_and only");
  run<25>("whole A and B slots by ds_read_b128 (B image swapped), the mix");
  return bad!=0;
}
