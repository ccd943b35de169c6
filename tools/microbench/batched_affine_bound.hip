// Upper bound for BATCHED-AFFINE bucket accumulation on gfx950, next to the XYZZ mixed addition the
// MSM uses.   hipcc --offload-arch=gfx950 -O3 -std=c++17 batched_affine_bound.hip -o batched_affine_bound && ./batched_affine_bound
//
// An affine addition costs  lambda = dy * inv, lambda^2, lambda * (x1 - x3)  = 2M + 1S  once 1 / (x2 - x1) is known, and
// Montgomery's trick supplies the inverses of K denominators for 3 (K - 1) products + ONE inversion.  The four kernels
// price the pieces separately (operands are synthetic: this measures the instruction mix, not a verified sum):
//   k_madd            the shipped xyzz_madd, register-only loop                      -> baseline additions / s
//   k_affine_free     5M + 1S + the add/sub glue per addition, K = 4 chains per lane in registers, the batch inverse
//                     replaced by a constant: batched affine with a FREE inversion and FREE prefix storage (upper bound)
//   k_affine_lds      the same with the K = 16 prefix products of a lane staged through LDS (36 B each: 147 KiB per
//                     256-lane workgroup, i.e. ONE workgroup per CU) -- the storage a real kernel needs
//   k_shared_inverse  what sharing one inversion per wave costs: butterfly exclusive product over 64 lanes (12 products)
//                     + safegcd on one lane + 1 product, per batch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../myzkp_amd/csrc/mzk_ec.h"
using namespace mzk;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef FqParams P;
__device__ __forceinline__ Fq synth(int s) { Fq x; for (int i = 0; i < 9; i++) x.l[i] = (u32)(s * 2654435761u + i * 40503u + 12345u) & MASK29; x.l[8] &= 0xfffff; return x; }
__device__ __forceinline__ u32 fold(const Fq& x) { u32 r = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) r ^= x.l[i]; return r; }

__global__ __launch_bounds__(256) void k_madd(u32* out, int iters) {
  u32 one[8] = {1,0,0,0,0,0,0,0}, two[8] = {2,0,0,0,0,0,0,0};
  Affine g; g.x = fe_reduce<P>(fe_to_mont<P>(fe_unpack<P>(one))); g.y = fe_reduce<P>(fe_to_mont<P>(fe_unpack<P>(two)));
  Xyzz acc = xyzz_dbl_affine(g);
  for (int k = 0; k < (int)(threadIdx.x & 7); k++) acc = xyzz_dbl(acc);
  for (int k = 0; k < iters; k++) acc = xyzz_madd(acc, g);
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(acc.X) ^ fold(acc.Y) ^ fold(acc.ZZ);
}
// one batched-affine step over K additions of a lane, inverse of the batch product supplied (free).  Written with
// macros over named registers: arrays indexed inside the unrolled loops stayed in scratch memory.
#define AFF_FWD(J, A)                                                              \
  { const Fq d = fe_carry<P>(fe_sub<P, 4>(x2, x1_##A));                             \
    if (LDS) { _Pragma("unroll") for (int i = 0; i < 9; i++) mine[(J) * 9 + i] = run.l[i]; } else pre_##J = run; \
    run = fe_mul<P>(run, d); }
#define AFF_BWD(J, A)                                                              \
  { Fq pre;                                                                         \
    if (LDS) { _Pragma("unroll") for (int i = 0; i < 9; i++) pre.l[i] = mine[(J) * 9 + i]; } else pre = pre_##J; \
    const Fq d = fe_carry<P>(fe_sub<P, 4>(x2, x1_##A));                             \
    const Fq dinv = fe_mul<P>(inv, pre);                                            \
    inv = fe_mul<P>(inv, d);                                                        \
    const Fq lam = fe_mul<P>(fe_carry<P>(fe_sub<P, 4>(y2, y1_##A)), dinv);          \
    const Fq l2 = fe_sqr<P>(lam);                                                   \
    const Fq x3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(l2, x1_##A), x2));    \
    const Fq y3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_mul<P>(lam, fe_carry<P>(fe_sub<P, 8>(x1_##A, x3))), y1_##A)); \
    x1_##A = x3; y1_##A = y3; }
template <int K, bool LDS>
__global__ __launch_bounds__(256) void k_affine(u32* out, int iters) {
  extern __shared__ u32 sh[];
  Fq x1_0 = synth(threadIdx.x * 8), x1_1 = synth(threadIdx.x * 8 + 1), x1_2 = synth(threadIdx.x * 8 + 2), x1_3 = synth(threadIdx.x * 8 + 3);
  Fq y1_0 = synth(threadIdx.x * 8 + 4), y1_1 = synth(threadIdx.x * 8 + 5), y1_2 = synth(threadIdx.x * 8 + 6), y1_3 = synth(threadIdx.x * 8 + 7);
  const Fq x2 = synth(7777), y2 = synth(8888), fake_inv = synth(9999);
  Fq pre_0, pre_1, pre_2, pre_3, pre_4, pre_5, pre_6, pre_7, pre_8, pre_9, pre_10, pre_11, pre_12, pre_13, pre_14, pre_15;
  u32* mine = sh + (size_t)threadIdx.x * K * 9;
  for (int it = 0; it < iters; it++) {
    Fq run = fe_one<P>();
    AFF_FWD(0, 0) AFF_FWD(1, 1) AFF_FWD(2, 2) AFF_FWD(3, 3)
    if (K >= 8) { AFF_FWD(4, 0) AFF_FWD(5, 1) AFF_FWD(6, 2) AFF_FWD(7, 3) }
    if (K >= 16) { AFF_FWD(8, 0) AFF_FWD(9, 1) AFF_FWD(10, 2) AFF_FWD(11, 3) AFF_FWD(12, 0) AFF_FWD(13, 1) AFF_FWD(14, 2) AFF_FWD(15, 3) }
    Fq inv = fe_mul<P>(run, fake_inv);              // stands in for the (shared) inversion
    if (K >= 16) { AFF_BWD(15, 3) AFF_BWD(14, 2) AFF_BWD(13, 1) AFF_BWD(12, 0) AFF_BWD(11, 3) AFF_BWD(10, 2) AFF_BWD(9, 1) AFF_BWD(8, 0) }
    if (K >= 8) { AFF_BWD(7, 3) AFF_BWD(6, 2) AFF_BWD(5, 1) AFF_BWD(4, 0) }
    AFF_BWD(3, 3) AFF_BWD(2, 2) AFF_BWD(1, 1) AFF_BWD(0, 0)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x1_0) ^ fold(y1_0) ^ fold(x1_1) ^ fold(y1_1) ^ fold(x1_2) ^ fold(y1_2) ^ fold(x1_3) ^ fold(y1_3);
}
__global__ __launch_bounds__(256) void k_shared_inverse(u32* out, int iters) {
  const int lane = threadIdx.x & 63;
  Fq t = synth(threadIdx.x + 17);
  u32 acc = 0;
  for (int it = 0; it < iters; it++) {
    Fq incl = t, excl = fe_one<P>();
    for (int m = 1; m < 64; m <<= 1) {              // butterfly: exclusive and inclusive products over the wave
      Fq other;
      for (int i = 0; i < 9; i++) other.l[i] = (u32)__shfl_xor((int)incl.l[i], m);
      excl = fe_mul<P>(excl, other);
      incl = fe_mul<P>(incl, other);
    }
    Fq inv = incl;
    if (lane == 0) inv = fe_inv_safegcd<P>(fe_reduce<P>(incl));
    for (int i = 0; i < 9; i++) inv.l[i] = (u32)__shfl((int)inv.l[i], 0);
    t = fe_mul<P>(inv, excl);                       // 1 / t_lane
    acc ^= fold(t);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <class F> static float time_it(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f, ms;
  for (int rep = 0; rep < 6; rep++) {               // clocks ramp: keep the best of six
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); const int ncu = pr.multiProcessorCount;
  u32* out; CK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4));
  CK(hipFuncSetAttribute((const void*)k_affine<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int iters = 128;
  for (int bpc : {1, 4}) {
    dim3 g(ncu * bpc), b(256);
    float ms = time_it([&] { hipLaunchKernelGGL(k_madd, g, b, 0, 0, out, iters * 4); });
    printf("xyzz_madd (shipped)                     blocks/CU=%d: %8.3f ms  %7.2f G additions/s\n", bpc, ms, (double)g.x * 256 * iters * 4 / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL((k_affine<4, false>), g, b, 0, 0, out, iters); });
    printf("batched affine, free inverse, K=4 regs  blocks/CU=%d: %8.3f ms  %7.2f G additions/s\n", bpc, ms, (double)g.x * 256 * iters * 4 / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL((k_affine<8, false>), g, b, 0, 0, out, iters / 2); });
    printf("batched affine, free inverse, K=8 regs  blocks/CU=%d: %8.3f ms  %7.2f G additions/s\n", bpc, ms, (double)g.x * 256 * (iters / 2) * 8 / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL(k_shared_inverse, g, b, 0, 0, out, 16); });
    printf("shared inversion per wave-batch         blocks/CU=%d: %8.3f ms  %7.3f us per batch per wave (all waves busy)\n", bpc, ms, ms * 1e3 / 16);
  }
  {
    dim3 g(ncu), b(256);                             // 147 KiB of prefix products: one workgroup per CU is all that fits
    float ms = time_it([&] { hipLaunchKernelGGL((k_affine<16, true>), g, b, 256 * 16 * 36, 0, out, iters / 4); });
    printf("batched affine, free inverse, K=16 LDS  blocks/CU=1: %8.3f ms  %7.2f G additions/s\n", ms, (double)g.x * 256 * (iters / 4) * 16 / ms / 1e6);
  }
  return 0;
}
