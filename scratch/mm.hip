#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
struct Fr { static constexpr int N=8;
 static constexpr u32 MOD[8]={0xf0000001u,0x43e1f593u,0x79b97091u,0x2833e848u,0x8181585du,0xb85045b6u,0xe131a029u,0x30644e72u};
 static constexpr u32 INV=0xefffffffu; };
template<class P> struct fe { u32 l[P::N]; };

template<class P> __device__ __forceinline__ fe<P> mont_mul(const fe<P>&a,const fe<P>&b){
  constexpr int N=P::N; u32 t[N+2];
  #pragma unroll
  for(int i=0;i<N+2;i++) t[i]=0;
  #pragma unroll
  for(int i=0;i<N;i++){
    u64 c=0;
    #pragma unroll
    for(int j=0;j<N;j++){ u64 acc=(u64)a.l[j]*b.l[i]+t[j]+c; t[j]=(u32)acc; c=acc>>32; }
    u64 s=(u64)t[N]+c; t[N]=(u32)s; t[N+1]=(u32)(s>>32);
    u32 m=t[0]*P::INV;
    u64 acc=(u64)m*P::MOD[0]+t[0]; c=acc>>32;
    #pragma unroll
    for(int j=1;j<N;j++){ acc=(u64)m*P::MOD[j]+t[j]+c; t[j-1]=(u32)acc; c=acc>>32; }
    s=(u64)t[N]+c; t[N-1]=(u32)s; t[N]=t[N+1]+(u32)(s>>32);
  }
  // conditional subtract
  fe<P> r; u32 br=0; u32 d[N];
  #pragma unroll
  for(int j=0;j<N;j++){ u64 x=(u64)t[j]-P::MOD[j]-br; d[j]=(u32)x; br=(x>>63)&1; }
  bool ge = (t[N]!=0) || (br==0);
  #pragma unroll
  for(int j=0;j<N;j++) r.l[j]= ge? d[j]:t[j];
  return r;
}
extern "C" __global__ void k_mul(const fe<Fr>* a,const fe<Fr>* b, fe<Fr>* o,int iters){
  int i=blockIdx.x*blockDim.x+threadIdx.x; fe<Fr> x=a[i],y=b[i];
  for(int k=0;k<iters;k++){ x=mont_mul<Fr>(x,y); }
  o[i]=x;
}
