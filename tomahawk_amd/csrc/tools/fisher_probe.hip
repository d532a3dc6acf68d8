// Dev tool: device lgamma / Fisher P against the host's (glibc) at large tables.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../hip/ld_math.hip.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__global__ void k_lg(const double* x, double* y, int n){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i]=lgamma(x[i]); }
__global__ void k_fi(const int* t, double* p, double* q, int n){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n){ p[i]=twk::d_fisher_two(twk::LFact{nullptr, 0}, t[4*i],t[4*i+1],t[4*i+2],t[4*i+3]);
  const int n1_=t[4*i]+t[4*i+1], n_1=t[4*i]+t[4*i+2], nn=t[4*i]+t[4*i+1]+t[4*i+2]+t[4*i+3]; q[i]=twk::d_hypergeo(twk::LFact{nullptr, 0}, t[4*i],n1_,n_1,nn);} }
static double h_lbinom(int n,int k){ if(k==0||n==k) return 0; return lgamma(n+1.0)-lgamma(k+1.0)-lgamma(n-k+1.0); }
int main(){
  const int n=12; double hx[n]={10.5,1e3+1,1e5+1,2e6+1,8388609.0,1.2e7+1,16777217.0,2e7+1,19583418.0,4e7+1,126360.0,4044197.0};
  double *dx,*dy; CK(hipMalloc(&dx,n*8)); CK(hipMalloc(&dy,n*8)); CK(hipMemcpy(dx,hx,n*8,hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_lg,dim3(1),dim3(64),0,0,dx,dy,n); double hy[n]; CK(hipMemcpy(hy,dy,n*8,hipMemcpyDeviceToHost));
  for(int i=0;i<n;++i){ double r=lgamma(hx[i]); printf("lgamma(%.1f): device %.17g host %.17g  diff %.3e (%.1f ulp)\n",hx[i],hy[i],r,hy[i]-r,(hy[i]-r)/(nextafter(r,1e300)-r)); }
  const int m=4; int ht[4*m]={15697165,126359,4044196,32616, 19583417,157964,158400,1295, 15696885,126725,4044476,32250, 1213,403,300,84};
  int* dt; double *dp,*dq; CK(hipMalloc(&dt,sizeof(ht))); CK(hipMalloc(&dp,m*8)); CK(hipMalloc(&dq,m*8)); CK(hipMemcpy(dt,ht,sizeof(ht),hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_fi,dim3(1),dim3(64),0,0,dt,dp,dq,m); double hp[m],hq[m]; CK(hipMemcpy(hp,dp,m*8,hipMemcpyDeviceToHost)); CK(hipMemcpy(hq,dq,m*8,hipMemcpyDeviceToHost));
  for(int i=0;i<m;++i){ const int* t=ht+4*i; const int n1_=t[0]+t[1],n_1=t[0]+t[2],nn=t[0]+t[1]+t[2]+t[3];
    double q=exp(h_lbinom(n1_,t[0])+h_lbinom(nn-n1_,n_1-t[0])-h_lbinom(nn,n_1));
    printf("table %d %d %d %d: device P %.17g  pmf(obs) device %.17g host %.17g (rel %.2e)\n",t[0],t[1],t[2],t[3],hp[i],hq[i],q,hq[i]/q-1); }
  return 0;
}
