#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../myzkp_amd/csrc/mzk_field.h"
using namespace mzk;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef FqParams P;
#define BAR(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ Fe<P> fe_mul_chain(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L; u32 m[L]; Fe<P> r; u64 col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) col += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) col += (u64)m[i] * P::P[k - i];
    m[k] = ((u32)col * P::N0) & MASK29;
    col += (u64)m[k] * P::P[0];
    col >>= W29;
    BAR(col);
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col += (u64)m[i] * P::P[k - i];
    r.l[k - L] = (u32)col & MASK29; col >>= W29;
    BAR(col);
  }
  r.l[L - 1] = (u32)col; return r;
}
#define MADV(col, x, y) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(col) : "v"(x), "v"(y) : "vcc")
#define MADS(col, x, c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(col) : "v"(x), "s"(c) : "vcc")
__device__ __forceinline__ Fe<P> fe_mul_asm(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L; u32 m[L]; Fe<P> r; u64 col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) MADV(col, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) MADS(col, m[i], P::P[k - i]);
    m[k] = ((u32)col * P::N0) & MASK29;
    MADS(col, m[k], P::P[0]);
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) MADV(col, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) MADS(col, m[i], P::P[k - i]);
    r.l[k - L] = (u32)col & MASK29; col >>= W29;
  }
  r.l[L - 1] = (u32)col; return r;
}
template<int V> __global__ void k_chain(u32* out, int iters, u32 seed) {
  Fe<P> x, y;
  for (int i = 0; i < P::L; i++) { x.l[i] = (seed * (i + 3) + threadIdx.x) & MASK29; y.l[i] = (seed * (i + 11)) & MASK29; }
  for (int k = 0; k < iters; k++) { if (V == 0) x = fe_mul<P>(x, y); else if (V == 1) x = fe_mul_chain(x, y); else x = fe_mul_asm(x, y); }
  for (int i = 0; i < P::L; i++) out[(blockIdx.x*blockDim.x + threadIdx.x) * P::L + i] = x.l[i];
}
int main() {
  u32* out; CK(hipMalloc(&out, (size_t)2048*256 * 9 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  for (int rep = 0; rep < 4; rep++) for (int v = 0; v < 3; v++) {
    int iters = 2000, blocks = 2048, threads = 256;
    if (v == 0) { CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_chain<0>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }
    else if (v == 1) { CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_chain<1>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }
    else { CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_chain<2>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("variant %d: %.1f Gmul/s\n", v, (double)blocks*threads*iters/ms/1e6);
  }
  {
    u32 *o0, *o2; CK(hipMalloc(&o0, 64*9*4)); CK(hipMalloc(&o2, 64*9*4));
    hipLaunchKernelGGL((k_chain<0>), dim3(1), dim3(64), 0, 0, o0, 50, 777u); hipLaunchKernelGGL((k_chain<2>), dim3(1), dim3(64), 0, 0, o2, 50, 777u);
    u32 h0[64*9], h2[64*9]; CK(hipMemcpy(h0, o0, sizeof h0, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, o2, sizeof h2, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 64*9; i++) bad += h0[i] != h2[i];
    printf("asm variant mismatches: %d\n", bad);
  }
  return 0;
}
