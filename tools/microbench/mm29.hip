#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
// Fr in 9x29-bit limbs
struct Fr29 { static constexpr int L=9; static constexpr int W=29; static constexpr u32 MASK=(1u<<29)-1;
  // p limbs (29-bit) and n0' = -p^-1 mod 2^29 are filled by host at init into constant memory for this probe
};
__constant__ u32 P29[9]; __constant__ u32 N0;
struct fe29 { u32 l[9]; };

__device__ __forceinline__ fe29 mont_mul29(const fe29&a,const fe29&b,const u32* __restrict__ p,u32 n0){
  constexpr int L=9; constexpr u32 MASK=(1u<<29)-1;
  u32 m[L]; fe29 r; u64 col=0;
  #pragma unroll
  for(int k=0;k<L;k++){
    #pragma unroll
    for(int i=0;i<=k;i++) col += (u64)a.l[i]*b.l[k-i];
    #pragma unroll
    for(int i=0;i<k;i++) col += (u64)m[i]*p[k-i];
    m[k]=((u32)col*n0)&MASK;
    col += (u64)m[k]*p[0];
    col >>= 29;
  }
  #pragma unroll
  for(int k=L;k<2*L-1;k++){
    #pragma unroll
    for(int i=k-L+1;i<L;i++) col += (u64)a.l[i]*b.l[k-i];
    #pragma unroll
    for(int i=k-L+1;i<L;i++) col += (u64)m[i]*p[k-i];
    r.l[k-L]=(u32)col&MASK; col>>=29;
  }
  r.l[L-1]=(u32)col;
  return r;
}
extern "C" __global__ void k_mul29(const fe29* a,const fe29* b, fe29* o,int iters){
  int i=blockIdx.x*blockDim.x+threadIdx.x; fe29 x=a[i],y=b[i];
  u32 p[9]; for(int j=0;j<9;j++) p[j]=P29[j]; u32 n0=N0;
  for(int k=0;k<iters;k++){ x=mont_mul29(x,y,p,n0); }
  o[i]=x;
}
// variant with compile-time constant modulus
__device__ __forceinline__ fe29 mont_mul29c(const fe29&a,const fe29&b){
  constexpr u32 p[9]={0x10000001,0x1f0fac9f,0x0e5c2450,0x07d090f3,0x1585d283,0x02db40c0,0x00a6e141,0x0e5c2634,0x0030644e};
  constexpr u32 n0=0x0fffffff;
  constexpr int L=9; constexpr u32 MASK=(1u<<29)-1;
  u32 m[L]; fe29 r; u64 col=0;
  #pragma unroll
  for(int k=0;k<L;k++){
    #pragma unroll
    for(int i=0;i<=k;i++) col += (u64)a.l[i]*b.l[k-i];
    #pragma unroll
    for(int i=0;i<k;i++) col += (u64)m[i]*p[k-i];
    m[k]=((u32)col*n0)&MASK;
    col += (u64)m[k]*p[0];
    col >>= 29;
  }
  #pragma unroll
  for(int k=L;k<2*L-1;k++){
    #pragma unroll
    for(int i=k-L+1;i<L;i++) col += (u64)a.l[i]*b.l[k-i];
    #pragma unroll
    for(int i=k-L+1;i<L;i++) col += (u64)m[i]*p[k-i];
    r.l[k-L]=(u32)col&MASK; col>>=29;
  }
  r.l[L-1]=(u32)col;
  return r;
}
extern "C" __global__ void k_mul29c(const fe29* a,const fe29* b, fe29* o,int iters){
  int i=blockIdx.x*blockDim.x+threadIdx.x; fe29 x=a[i],y=b[i];
  for(int k=0;k<iters;k++){ x=mont_mul29c(x,y); }
  o[i]=x;
}
int main(){
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr,0)); int ncu=pr.multiProcessorCount;
  u32 hp[9]={0x10000001,0x1f0fac9f,0x0e5c2450,0x07d090f3,0x1585d283,0x02db40c0,0x00a6e141,0x0e5c2634,0x0030644e}; u32 hn0=0x0fffffff;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(P29),hp,36)); CK(hipMemcpyToSymbol(HIP_SYMBOL(N0),&hn0,4));
  int n=ncu*8*256; fe29*a,*b,*o; CK(hipMalloc(&a,n*36)); CK(hipMalloc(&b,n*36)); CK(hipMalloc(&o,n*36));
  CK(hipMemset(a,0x0a,n*36)); CK(hipMemset(b,0x07,n*36));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int v=0;v<2;v++) for(int bpc: {1,2,4,8}){ int iters=400; dim3 g(ncu*bpc), bl(256);
    auto kk = v? k_mul29c : k_mul29;
    hipLaunchKernelGGL(kk,g,bl,0,0,a,b,o,4); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(kk,g,bl,0,0,a,b,o,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); double muls=(double)iters*g.x*bl.x;
    printf("mont_mul Fr 9x29 FIPS %s: blocks/CU=%d %8.3f ms  %8.2f Gmul/s\n",v?"constmod":"regmod",bpc,ms,muls/(ms*1e-3)/1e9);
  }
  return 0;
}
