#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../myzkp_amd/csrc/mzk_ec.h"
using namespace mzk;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef FqParams P;
__global__ void k_inv(u32* out, u32 seed, int active) {
  if ((int)threadIdx.x >= active) return;
  Fe<P> x; for (int i = 0; i < P::L; i++) x.l[i] = (seed * (i + 3) + threadIdx.x) & MASK29;
  x = fe_inv<P>(x);
  for (int i = 0; i < P::L; i++) out[threadIdx.x * 9 + i] = x.l[i];
}
__global__ void k_dbl(u32* out, u32 seed, int n, int active) {
  if ((int)threadIdx.x >= active) return;
  Affine a; for (int i = 0; i < P::L; i++) { a.x.l[i] = (seed * (i + 3) + threadIdx.x) & MASK29; a.y.l[i] = (seed * (i + 7)) & MASK29; }
  a.x.l[8] &= 0xfffff; a.y.l[8] &= 0xfffff;
  Xyzz p = xyzz_from_affine(a);
  for (int k = 0; k < n; k++) p = xyzz_dbl(p);
  for (int i = 0; i < P::L; i++) out[threadIdx.x * 9 + i] = p.X.l[i] ^ p.Y.l[i] ^ p.ZZ.l[i] ^ p.ZZZ.l[i];
}
__global__ void k_add(u32* out, u32 seed, int n, int active) {
  if ((int)threadIdx.x >= active) return;
  Affine a; for (int i = 0; i < P::L; i++) { a.x.l[i] = (seed * (i + 3) + threadIdx.x) & MASK29; a.y.l[i] = (seed * (i + 7)) & MASK29; }
  a.x.l[8] &= 0xfffff; a.y.l[8] &= 0xfffff;
  Xyzz p = xyzz_from_affine(a), q = xyzz_dbl(p);
  for (int k = 0; k < n; k++) p = xyzz_add(p, q);
  for (int i = 0; i < P::L; i++) out[threadIdx.x * 9 + i] = p.X.l[i] ^ p.Y.l[i] ^ p.ZZ.l[i] ^ p.ZZZ.l[i];
}
int main() {
  u32* out; CK(hipMalloc(&out, 64 * 9 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  for (int active : {1, 64}) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_inv, dim3(1), dim3(64), 0, 0, out, 12345u, active); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); printf("fe_inv   active=%2d: %.1f us (%.3f us per op, 381 ops)\n", active, ms * 1e3, ms * 1e3 / 381);
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_dbl, dim3(1), dim3(64), 0, 0, out, 12345u, 240, active); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); printf("240 dbl  active=%2d: %.1f us (%.3f us per dbl = %.3f per mul-equiv)\n", active, ms * 1e3, ms * 1e3 / 240, ms*1e3/240/9);
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, 0, out, 12345u, 100, active); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); printf("100 add  active=%2d: %.1f us (%.3f us per add = %.3f per mul-equiv)\n", active, ms * 1e3, ms * 1e3 / 100, ms*1e3/100/14);
    }
  }
  return 0;
}
