#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../myzkp_amd/csrc/mzk_ec.h"
#include "../myzkp_amd/csrc/mzk_coop.h"
using namespace mzk;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
__device__ Affine gen() {
  u32 one[8] = {1,0,0,0,0,0,0,0}, two[8] = {2,0,0,0,0,0,0,0};
  Affine g; g.x = fe_reduce<FqParams>(fe_to_mont<FqParams>(fe_unpack<FqParams>(one))); g.y = fe_reduce<FqParams>(fe_to_mont<FqParams>(fe_unpack<FqParams>(two)));
  return g;
}
__global__ void k_test(int n, u32* out) {
  const int lane = threadIdx.x & 3;
  Xyzz a = xyzz_from_affine(gen()), b = a;
  const Xyzz g3 = xyzz_madd(xyzz_dbl_affine(gen()), gen());
  for (int k = 0; k < n; k++) { a = xyzz_dbl(a); b = xyzz_dbl_quad(b, lane); a = xyzz_add(a, g3); b = xyzz_add_quad(b, g3, lane); }
  Fq l = fe_reduce<FqParams>(fe_mul<FqParams>(a.X, b.ZZ)), r = fe_reduce<FqParams>(fe_mul<FqParams>(b.X, a.ZZ));
  Fq l2 = fe_reduce<FqParams>(fe_mul<FqParams>(a.Y, b.ZZZ)), r2 = fe_reduce<FqParams>(fe_mul<FqParams>(b.Y, a.ZZZ));
  out[threadIdx.x] = (fe_eq_canon<FqParams>(l, r) && fe_eq_canon<FqParams>(l2, r2) && !xyzz_is_inf(b)) ? 1 : 0;
}
int main() {
  u32* out; CK(hipMalloc(&out, 1024));
  for (int threads : {4, 64, 256}) for (int n : {1, 2, 8}) {
    CK(hipMemset(out, 0, 1024));
    hipLaunchKernelGGL(k_test, dim3(1), dim3(threads), 0, 0, n, out);
    CK(hipDeviceSynchronize());
    u32 h[256]; CK(hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost));
    int ok = 0; for (int i = 0; i < threads; i++) ok += h[i];
    printf("threads %d n %d: %d/%d ok\n", threads, n, ok, threads);
  }
  return 0;
}
