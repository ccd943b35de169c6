// single-wave dependent-chain latency of fe_mul variants (scratch)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../myzkp_amd/csrc/mzk_field.h"
using namespace mzk;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef FqParams P;
// variant: two accumulators per column
__device__ __forceinline__ Fe<P> fe_mul2(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L; u32 m[L]; Fe<P> r; u64 col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    u64 c2 = 0;
#pragma unroll
    for (int i = 0; i <= k; i++) col += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) c2 += (u64)m[i] * P::P[k - i];
    col += c2;
    m[k] = ((u32)col * P::N0) & MASK29;
    col += (u64)m[k] * P::P[0];
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
    u64 c2 = 0;
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) c2 += (u64)m[i] * P::P[k - i];
    col += c2;
    r.l[k - L] = (u32)col & MASK29; col >>= W29;
  }
  r.l[L - 1] = (u32)col; return r;
}
// variant: separated product then reduction (product columns are all independent)
__device__ __forceinline__ Fe<P> fe_mul3(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L; u64 t[2 * L - 1];
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    u64 c = 0;
#pragma unroll
    for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); i++) c += (u64)a.l[i] * b.l[k - i];
    t[k] = c;
  }
  u32 m[L]; Fe<P> r; u64 col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
    u64 c2 = t[k];
#pragma unroll
    for (int i = 0; i < k; i++) c2 += (u64)m[i] * P::P[k - i];
    col += c2;
    m[k] = ((u32)col * P::N0) & MASK29;
    col += (u64)m[k] * P::P[0];
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
    u64 c2 = t[k];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) c2 += (u64)m[i] * P::P[k - i];
    col += c2;
    r.l[k - L] = (u32)col & MASK29; col >>= W29;
  }
  r.l[L - 1] = (u32)col; return r;
}
template<int V> __global__ void k_chain(u32* out, int iters, u32 seed) {
  Fe<P> x, y;
  for (int i = 0; i < P::L; i++) { x.l[i] = (seed * (i + 3) + threadIdx.x) & MASK29; y.l[i] = (seed * (i + 11)) & MASK29; }
  for (int k = 0; k < iters; k++) {
    if (V == 0) x = fe_mul<P>(x, y); else if (V == 1) x = fe_mul2(x, y); else if (V==2) x = fe_mul3(x, y); else x = fe_sqr<P>(x);
  }
  for (int i = 0; i < P::L; i++) out[threadIdx.x * P::L + i + blockIdx.x*blockDim.x*P::L] = x.l[i];
}
template<int V> int run(const char* name, int blocks, int threads) {
  u32* out; CK(hipMalloc(&out, (size_t)blocks*threads * 9 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 2000;
  hipLaunchKernelGGL((k_chain<V>), dim3(blocks), dim3(threads), 0, 0, out, 10, 12345u); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_chain<V>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s blocks=%5d threads=%4d: %.3f us per op in chain; %.1f Gop/s\n", name, blocks, threads, ms * 1e3 / iters, (double)blocks*threads*iters/ms/1e6);
  return 0;
}
int main() {
  for (int cfg = 0; cfg < 3; cfg++) {
    int blocks = cfg == 0 ? 1 : (cfg == 1 ? 256 : 256*8), threads = cfg == 0 ? 64 : 256;
    run<0>("fe_mul (1 chain)", blocks, threads); run<1>("fe_mul2 (2 chains)", blocks, threads); run<2>("fe_mul3 (product first)", blocks, threads); run<3>("fe_sqr", blocks, threads);
  }
  return 0;
}
