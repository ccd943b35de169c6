// Instruction-mix microbenchmark for gfx950 (scratch tool, not product): rates of the NON-MAD instructions of the
// Montgomery product (64-bit shift, alignbit, and, mul_lo, 64-bit add) alone and interleaved with v_mad_u64_u32, and
// fe_mul variants.  hipcc --offload-arch=gfx950 -O3 -I../../myzkp_amd/csrc ubench2.hip -o ubench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "mzk_field.h"
#include "mzk_field_asm.h"
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

// A: 64-bit accumulators (8), B: 32-bit accumulators (8).  NA / NB instructions of each kind per inner round.
#define KMIX(NAME, NA, ASMA, NB, ASMB) \
extern "C" __global__ void NAME(u64* out, int iters, u32 sa, u32 sb){ \
  u64 acc[8]; u32 bcc[8]; u32 a=threadIdx.x+sa; u32 b=threadIdx.x*7+sb; u64 b64=b; \
  for(int i=0;i<8;i++){ acc[i]=i+sa+threadIdx.x; bcc[i]=acc[i]*3; } \
  for(int k=0;k<iters;k++){ _Pragma("unroll") for(int r=0;r<4;r++){ _Pragma("unroll") for(int i=0;i<8;i++){ \
      if (i < NA) asm volatile(ASMA : "+v"(acc[i]) : "v"(a),"v"(b),"v"(b64) : "vcc"); \
      if (i < NB) asm volatile(ASMB : "+v"(bcc[i]) : "v"(a),"v"(b),"v"(acc[(i+4)&7]) : "vcc"); } } } \
  u64 s=0; for(int i=0;i<8;i++) s^=acc[i]^bcc[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }

KMIX(k_mad,        8, "v_mad_u64_u32 %0, vcc, %1, %2, %0", 0, "")
KMIX(k_shr64,      8, "v_lshrrev_b64 %0, 29, %0", 0, "")
KMIX(k_lshladd,    8, "v_lshl_add_u64 %0, %0, 0, %3", 0, "")
KMIX(k_and,        0, "", 8, "v_and_b32 %0, 0x1fffffff, %0")
KMIX(k_alignbit,   0, "", 8, "v_alignbit_b32 %0, %0, %2, 29")
KMIX(k_add3,       0, "", 8, "v_add3_u32 %0, %0, %1, %2")
KMIX(k_mullo,      0, "", 8, "v_mul_lo_u32 %0, %0, %2")
KMIX(k_cndmask,    0, "", 8, "v_cndmask_b32 %0, %0, %2, vcc")
KMIX(k_mad_and,    8, "v_mad_u64_u32 %0, vcc, %1, %2, %0", 8, "v_and_b32 %0, 0x1fffffff, %0")
KMIX(k_mad_and4,   8, "v_mad_u64_u32 %0, vcc, %1, %2, %0", 4, "v_and_b32 %0, 0x1fffffff, %0")
KMIX(k_mad_mullo,  8, "v_mad_u64_u32 %0, vcc, %1, %2, %0", 8, "v_mul_lo_u32 %0, %0, %2")
KMIX(k_mad_shr,    8, "v_mad_u64_u32 %0, vcc, %1, %2, %0", 0, "")   /* placeholder, see k_mad_shr2 */
extern "C" __global__ void k_mad_shr2(u64* out, int iters, u32 sa, u32 sb){
  u64 acc[8], c2[8]; u32 a=threadIdx.x+sa; u32 b=threadIdx.x*7+sb;
  for(int i=0;i<8;i++){ acc[i]=i+sa+threadIdx.x; c2[i]=acc[i]*5; }
  for(int k=0;k<iters;k++){
#pragma unroll
    for(int r=0;r<4;r++){
#pragma unroll
      for(int i=0;i<8;i++){
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a),"v"(b) : "vcc");
        asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(c2[i]));
      } } }
  u64 s=0; for(int i=0;i<8;i++) s^=acc[i]^c2[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }

// ---- fe_mul variants --------------------------------------------------------------------------------------------
using namespace mzk;
template <class P> __device__ __forceinline__ u64 mad_asm(u32 a, u32 b, u64 c) {
  u64 r; asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "vcc"); return r;
}
// V1: the carry of the previous column is the addend of the column's FIRST multiply-add (one asm per column)
template <class P> __device__ __forceinline__ Fe<P> fe_mul_v1(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L;
  u32 m[L]; Fe<P> r; u64 col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    col = mad_asm<P>(a.l[0], b.l[k], col);
#pragma unroll
    for (int i = 1; i <= k; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = 0; i < k; i++) col = mzk_mad(m[i], P::P[k - i], col);
    m[k] = ((u32)col * P::N0) & MASK29;
    col = mzk_mad(m[k], P::P[0], col);
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
    col = mad_asm<P>(a.l[k - L + 1], b.l[L - 1], col);
#pragma unroll
    for (int i = k - L + 2; i < L; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col = mzk_mad(m[i], P::P[k - i], col);
    r.l[k - L] = (u32)col & MASK29;
    col >>= W29;
  }
  r.l[L - 1] = (u32)col;
  return r;
}
// V2: shipped order, 64-bit shift written as alignbit + 32-bit shift
__device__ __forceinline__ u64 shr29(u64 c) {
  const u32 lo = (u32)c, hi = (u32)(c >> 32);
  return ((u64)(hi >> 29) << 32) | __builtin_amdgcn_alignbit(hi, lo, 29);
}
template <class P> __device__ __forceinline__ Fe<P> fe_mul_v2(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L;
  u32 m[L]; Fe<P> r; u64 col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = 0; i < k; i++) col = mzk_mad(m[i], P::P[k - i], col);
    m[k] = ((u32)col * P::N0) & MASK29;
    col = mzk_mad(m[k], P::P[0], col);
    col = shr29(col);
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col = mzk_mad(m[i], P::P[k - i], col);
    r.l[k - L] = (u32)col & MASK29;
    col = shr29(col);
  }
  r.l[L - 1] = (u32)col;
  return r;
}
// V3: product first (17 independent columns, no carries), then one carry/reduction sweep: separates the two halves so
// that the 81 product MADs have no dependence on the m's
template <class P> __device__ __forceinline__ Fe<P> fe_mul_v3(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L;
  u64 t[2 * L - 1];
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) { const int j = k - i; if (j >= 0 && j < L) c = mzk_mad(a.l[i], b.l[j], c); }
    t[k] = c;
  }
  u32 m[L]; Fe<P> r; u64 col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    col += t[k];
#pragma unroll
    for (int i = 0; i < k; i++) col = mzk_mad(m[i], P::P[k - i], col);
    m[k] = ((u32)col * P::N0) & MASK29;
    col = mzk_mad(m[k], P::P[0], col);
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
    col += t[k];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col = mzk_mad(m[i], P::P[k - i], col);
    r.l[k - L] = (u32)col & MASK29;
    col >>= W29;
  }
  r.l[L - 1] = (u32)col;
  return r;
}
template <int V> __global__ __launch_bounds__(256) void k_femul(const u32* in, u32* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fe<FqParams> x, y;
  for (int i = 0; i < 9; i++) { x.l[i] = in[t * 18 + i] & MASK29; y.l[i] = in[t * 18 + 9 + i] & MASK29; }
  for (int k = 0; k < iters; k++) {
    Fe<FqParams> z;
    if (V == 0) z = fe_mul<FqParams>(x, y);
    if (V == 1) z = fe_mul_v1<FqParams>(x, y);
    if (V == 2) z = fe_mul_v2<FqParams>(x, y);
    if (V == 3) z = fe_mul_v3<FqParams>(x, y);
    if (V == 4) z = fe_sqr<FqParams>(x);
    if (V == 5) z = FeAsm<FqParams>::mul(x, y);
    if (V == 6) z = FeAsm<FqParams>::sqr(x);
    if (V == 7) z = fe_mul_add2<FqParams>(x, y, y, x);
    if (V == 8) z = FeAsm<FqParams>::mul_add2(x, y, y, x);
    y = x; x = z;
  }
  for (int i = 0; i < 9; i++) out[t * 9 + i] = x.l[i];
}

typedef void (*kern_t)(u64*,int,u32,u32);
struct Case{ const char* name; kern_t k; int ops_per_iter; };
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  int ncu=p.multiProcessorCount;
  u64* out; CK(hipMalloc(&out, (size_t)ncu*64*256*8*2));
  std::vector<Case> cases={{"v_mad_u64_u32 x8",k_mad,32},{"v_lshrrev_b64 x8",k_shr64,32},{"v_lshl_add_u64 x8",k_lshladd,32},{"v_and_b32 x8",k_and,32},
    {"v_alignbit_b32 x8",k_alignbit,32},{"v_add3_u32 x8",k_add3,32},{"v_mul_lo_u32 x8",k_mullo,32},{"v_cndmask_b32 x8",k_cndmask,32},
    {"mad x8 + and x8",k_mad_and,32},{"mad x8 + and x4",k_mad_and4,32},{"mad x8 + mul_lo x8",k_mad_mullo,32},{"mad x8 + shr64 x8",k_mad_shr2,32}};
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int bpc : {1,4}) {
   printf("--- blocks/CU=%d x 256 threads (waves/SIMD=%d); cycles = SIMD cycles per wave-instruction GROUP (one A + its B's)\n",bpc,bpc);
   for(auto&c:cases){
    int iters=2000; dim3 g(ncu*bpc), b(256);
    hipLaunchKernelGGL(c.k,g,b,0,0,out,10,1u,2u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(c.k,g,b,0,0,out,iters,1u,2u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    // SIMD cycles per inner slot: time * clock / (iters * 32 slots * waves per SIMD)
    double cyc = ms*1e-3*2.4e9/((double)iters*32*bpc);
    printf("%-22s %8.3f ms   %6.2f cycles per slot\n",c.name,ms,cyc);
   }
  }
  {
    int n=ncu*4*256; u32 *in,*o; CK(hipMalloc(&in,(size_t)n*18*4)); CK(hipMalloc(&o,(size_t)n*9*4));
    CK(hipMemset(in,0x5a,(size_t)n*18*4));
    const char* names[9]={"fe_mul shipped","fe_mul v1 carry-as-addend","fe_mul v2 alignbit shift","fe_mul v3 product-then-reduce","fe_sqr shipped",
                          "fe_mul ASM block","fe_sqr ASM block","fe_mul_add2 shipped","fe_mul_add2 ASM block"};
    u32* o2; CK(hipMalloc(&o2,(size_t)n*9*4));
    std::vector<u32> h1((size_t)n*9), h2((size_t)n*9);
    {  // same results?
      dim3 g(ncu*4), bl(256);
      hipLaunchKernelGGL(k_femul<0>,g,bl,0,0,in,o,37); hipLaunchKernelGGL(k_femul<5>,g,bl,0,0,in,o2,37); CK(hipDeviceSynchronize());
      CK(hipMemcpy(h1.data(),o,h1.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(),o2,h2.size()*4,hipMemcpyDeviceToHost));
      printf("asm mul == C++ mul: %s\n", h1==h2 ? "yes" : "NO");
      hipLaunchKernelGGL(k_femul<4>,g,bl,0,0,in,o,37); hipLaunchKernelGGL(k_femul<6>,g,bl,0,0,in,o2,37); CK(hipDeviceSynchronize());
      CK(hipMemcpy(h1.data(),o,h1.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(),o2,h2.size()*4,hipMemcpyDeviceToHost));
      printf("asm sqr == C++ sqr: %s\n", h1==h2 ? "yes" : "NO");
      hipLaunchKernelGGL(k_femul<7>,g,bl,0,0,in,o,37); hipLaunchKernelGGL(k_femul<8>,g,bl,0,0,in,o2,37); CK(hipDeviceSynchronize());
      CK(hipMemcpy(h1.data(),o,h1.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(),o2,h2.size()*4,hipMemcpyDeviceToHost));
      printf("asm mul_add2 == C++ mul_add2: %s\n", h1==h2 ? "yes" : "NO");
    }
    for(int bpc : {1,2,3,4}) for(int v=0;v<9;v++){ if (v>=1 && v<=3) continue;
      int iters=2000; dim3 g(ncu*bpc), bl(256);
      auto launch=[&](int it){ switch(v){case 0: hipLaunchKernelGGL(k_femul<0>,g,bl,0,0,in,o,it); break; case 1: hipLaunchKernelGGL(k_femul<1>,g,bl,0,0,in,o,it); break;
        case 2: hipLaunchKernelGGL(k_femul<2>,g,bl,0,0,in,o,it); break; case 3: hipLaunchKernelGGL(k_femul<3>,g,bl,0,0,in,o,it); break; case 4: hipLaunchKernelGGL(k_femul<4>,g,bl,0,0,in,o,it); break;
        case 5: hipLaunchKernelGGL(k_femul<5>,g,bl,0,0,in,o,it); break; case 6: hipLaunchKernelGGL(k_femul<6>,g,bl,0,0,in,o,it); break; case 7: hipLaunchKernelGGL(k_femul<7>,g,bl,0,0,in,o,it); break;
        default: hipLaunchKernelGGL(k_femul<8>,g,bl,0,0,in,o,it);} };
      launch(4); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); double muls=(double)iters*g.x*bl.x;
      printf("%-32s waves/SIMD=%d %8.3f ms  %8.2f Gmul/s  %7.1f SIMD cycles per wave-mul\n",names[v],bpc,ms,muls/(ms*1e-3)/1e9, ms*1e-3*2.4e9/((double)iters*bpc));
    }
  }
  return 0;
}
