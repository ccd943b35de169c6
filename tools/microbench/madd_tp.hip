#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../myzkp_amd/csrc/mzk_ec.h"
using namespace mzk;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef FqParams P;
// G = (1,2) in Montgomery form built on device
__global__ __launch_bounds__(256) void k_madd(u32* out, int iters) {
  u32 one[8] = {1,0,0,0,0,0,0,0}, two[8] = {2,0,0,0,0,0,0,0};
  Affine g; g.x = fe_reduce<P>(fe_to_mont<P>(fe_unpack<P>(one))); g.y = fe_reduce<P>(fe_to_mont<P>(fe_unpack<P>(two)));
  Xyzz acc = xyzz_dbl_affine(g);
  for (int k = 0; k < (int)(threadIdx.x & 7); k++) acc = xyzz_dbl(acc);   // distinct starting points
  for (int k = 0; k < iters; k++) acc = xyzz_madd(acc, g);
  u32 w[32]; xyzz_store(acc, w);
  u32 x = 0; for (int i = 0; i < 32; i++) x ^= w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_mul10(u32* out, int iters) {
  Fe<P> x, y; for (int i = 0; i < 9; i++) { x.l[i] = (threadIdx.x * 77 + i * 1234567) & MASK29; y.l[i] = (i * 7654321 + 99) & MASK29; }
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int j = 0; j < 8; j++) x = fe_mul<P>(x, y);
    x = fe_sqr<P>(x); x = fe_sqr<P>(x);
  }
  u32 r = 0; for (int i = 0; i < 9; i++) r ^= x.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); int ncu = pr.multiProcessorCount;
  u32* out; CK(hipMalloc(&out, (size_t)ncu * 16 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  for (int bpc : {1, 2, 4}) {
    int iters = 256; dim3 g(ncu * bpc), b(256);
    hipLaunchKernelGGL(k_madd, g, b, 0, 0, out, 4); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_madd, g, b, 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("xyzz_madd  blocks/CU=%d: %.3f ms, %.2f G madd/s\n", bpc, ms, (double)g.x * 256 * iters / ms / 1e6);
    hipLaunchKernelGGL(k_mul10, g, b, 0, 0, out, 4); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mul10, g, b, 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("8M+2S bare blocks/CU=%d: %.3f ms, %.2f G (8M+2S)/s\n", bpc, ms, (double)g.x * 256 * iters / ms / 1e6);
  }
  return 0;
}
