// Instruction-throughput microbenchmark for gfx950 integer paths (scratch tool, not product).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)


#define K32(NAME, ASM) \
extern "C" __global__ void NAME(u64* out, int iters, u32 sa, u32 sb){ \
  u32 acc[8]; u32 a=threadIdx.x+sa; u32 b=threadIdx.x*7+sb; for(int i=0;i<8;i++) acc[i]=i+sa+threadIdx.x; \
  for(int k=0;k<iters;k++){ _Pragma("unroll") for(int r=0;r<4;r++){ _Pragma("unroll") for(int i=0;i<8;i++){ asm volatile(ASM : "+v"(acc[i]) : "v"(a),"v"(b) : "vcc"); } } } \
  u64 s=0; for(int i=0;i<8;i++) s^=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
#define K64(NAME, ASM) \
extern "C" __global__ void NAME(u64* out, int iters, u32 sa, u32 sb){ \
  u64 acc[8]; u32 a=threadIdx.x+sa; u32 b=threadIdx.x*7+sb; u64 b64=b; for(int i=0;i<8;i++) acc[i]=i+sa+threadIdx.x; \
  for(int k=0;k<iters;k++){ _Pragma("unroll") for(int r=0;r<4;r++){ _Pragma("unroll") for(int i=0;i<8;i++){ asm volatile(ASM : "+v"(acc[i]) : "v"(a),"v"(b),"v"(b64) : "vcc"); } } } \
  u64 s=0; for(int i=0;i<8;i++) s^=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
#define KF64(NAME, ASM) \
extern "C" __global__ void NAME(u64* out, int iters, u32 sa, u32 sb){ \
  double acc[8]; double a=1.0000001+threadIdx.x*1e-9+sa; double b=1e-9*sb; for(int i=0;i<8;i++) acc[i]=i; \
  for(int k=0;k<iters;k++){ _Pragma("unroll") for(int r=0;r<4;r++){ _Pragma("unroll") for(int i=0;i<8;i++){ asm volatile(ASM : "+v"(acc[i]) : "v"(a),"v"(b)); } } } \
  double s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u64)s; }
#define KF32(NAME, ASM) \
extern "C" __global__ void NAME(u64* out, int iters, u32 sa, u32 sb){ \
  float acc[8]; float a=1.0000001f+threadIdx.x*1e-9f+sa; float b=1e-9f*sb; for(int i=0;i<8;i++) acc[i]=i; \
  for(int k=0;k<iters;k++){ _Pragma("unroll") for(int r=0;r<4;r++){ _Pragma("unroll") for(int i=0;i<8;i++){ asm volatile(ASM : "+v"(acc[i]) : "v"(a),"v"(b)); } } } \
  float s=0; for(int i=0;i<8;i++) s+=acc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u64)s; }
K64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
K32(k_mullo, "v_mul_lo_u32 %0, %0, %2")
K32(k_mulhi, "v_mul_hi_u32 %0, %0, %2")
K32(k_mad24, "v_mad_u32_u24 %0, %1, %2, %0")
K32(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %2")
K64(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %3")
K32(k_add32, "v_add_u32 %0, %0, %2")
K32(k_addc, "v_addc_co_u32 %0, vcc, %0, %2, vcc")
K32(k_mov, "v_mov_b32 %0, %2")
K32(k_mad_i32_i24, "v_mad_u32_u16 %0, %1, %2, %0")
KF64(k_fma64, "v_fma_f64 %0, %1, %0, %2")
KF32(k_fma32, "v_fma_f32 %0, %1, %0, %2")
// dependent chain latency of v_mad_u64_u32 (1 accumulator)
extern "C" __global__ void k_mad64_dep(u64* out,int iters,u32 sa,u32 sb){
  u64 acc=sa; u32 a=threadIdx.x+sa,b=threadIdx.x*7+sb;
  for(int k=0;k<iters;k++){
    #pragma unroll
    for(int i=0;i<32;i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a),"v"(b) : "vcc");
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=acc;
}

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

typedef void (*kern_t)(u64*,int,u32,u32);
struct Case{ const char* name; kern_t k; int ops_per_iter; };

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs=%d clock=%d kHz\n",p.name,p.multiProcessorCount,p.clockRate);
  int ncu=p.multiProcessorCount;
  u64* out; CK(hipMalloc(&out, (size_t)ncu*64*256*8));
  std::vector<Case> cases={{"v_mad_u64_u32",k_mad64,32},{"v_mul_lo_u32",k_mullo,32},{"v_mul_hi_u32",k_mulhi,32},{"v_mad_u32_u24",k_mad24,32},
    {"v_mul_hi_u32_u24",k_mulhi24,32},{"v_lshl_add_u64",k_lshladd64,32},{"v_add_u32",k_add32,32},{"v_addc_co_u32",k_addc,32},{"v_mov_b32",k_mov,32},
    {"v_fma_f64",k_fma64,32},{"v_fma_f32",k_fma32,32},{"v_mad_u32_u16",k_mad_i32_i24,32},{"v_mad_u64_u32(dep)",k_mad64_dep,32}};
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int wpb : {4}) for(int bpc : {1,2,4,8}) {
   printf("--- blocks/CU=%d, waves/block=%d (waves/SIMD=%d)\n",bpc,wpb,bpc*wpb/4);
   for(auto&c:cases){
    int iters=2000; dim3 g(ncu*bpc), b(64*wpb);
    hipLaunchKernelGGL(c.k,g,b,0,0,out,10,1u,2u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(c.k,g,b,0,0,out,iters,1u,2u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    double ops=(double)iters*c.ops_per_iter*g.x*b.x; // lane-ops
    double rate=ops/(ms*1e-3);
    printf("%-22s %8.3f ms  %8.2f Tlane-ops/s  = %6.2f lane-ops/clk/CU @2.4GHz\n",c.name,ms,rate/1e12,rate/ncu/2.4e9);
   }
  }
  {
    int n=ncu*8*256; fe<Fr>*a,*b,*o; CK(hipMalloc(&a,n*32)); CK(hipMalloc(&b,n*32)); CK(hipMalloc(&o,n*32));
    CK(hipMemset(a,0x5a,n*32)); CK(hipMemset(b,0x17,n*32));
    for(int bpc: {1,2,4,8}){ int iters=400; dim3 g(ncu*bpc), bl(256);
      hipLaunchKernelGGL(k_mul,g,bl,0,0,a,b,o,4); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mul,g,bl,0,0,a,b,o,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); double muls=(double)iters*g.x*bl.x;
      printf("mont_mul Fr 8x32 CIOS: blocks/CU=%d %8.3f ms  %8.2f Gmul/s\n",bpc,ms,muls/(ms*1e-3)/1e9);
    }
  }
  return 0;
}
