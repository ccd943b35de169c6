#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../myzkp_amd/csrc/mzk_ec.h"
#include "../myzkp_amd/csrc/mzk_coop.h"
using namespace mzk;
__device__ Affine gen() {
  u32 one[8] = {1,0,0,0,0,0,0,0}, two[8] = {2,0,0,0,0,0,0,0};
  Affine g; g.x = fe_reduce<FqParams>(fe_to_mont<FqParams>(fe_unpack<FqParams>(one))); g.y = fe_reduce<FqParams>(fe_to_mont<FqParams>(fe_unpack<FqParams>(two)));
  return g;
}
__global__ void k_test(u32* out) {
  const int lane = threadIdx.x & 3;
  Xyzz b = xyzz_from_affine(gen());
  b = xyzz_dbl_quad(b, lane);
  u32 w[32];
  fe_pack<FqParams>(fe_reduce<FqParams>(b.X), w); fe_pack<FqParams>(fe_reduce<FqParams>(b.Y), w + 8);
  fe_pack<FqParams>(fe_reduce<FqParams>(b.ZZ), w + 16); fe_pack<FqParams>(fe_reduce<FqParams>(b.ZZZ), w + 24);
  for (int k = 0; k < 32; k++) out[threadIdx.x * 32 + k] = w[k];
}
int main() {
  u32* out; hipMalloc(&out, 4 * 128);
  hipLaunchKernelGGL(k_test, dim3(1), dim3(4), 0, 0, out);
  hipDeviceSynchronize();
  u32 h[128]; hipMemcpy(h, out, 512, hipMemcpyDeviceToHost);
  const char* nm[4] = {"X", "Y", "ZZ", "ZZZ"};
  for (int c = 0; c < 4; c++) { for (int l = 0; l < 4; l++) printf("%s lane%d: %08x %08x .. %08x\n", nm[c], l, h[l*32+c*8], h[l*32+c*8+1], h[l*32+c*8+7]); }
  return 0;
}
