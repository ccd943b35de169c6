#include <hip/hip_runtime.h>
#include "../myzkp_amd/csrc/mzk_field.h"
using namespace mzk;
typedef FqParams P;
extern "C" __global__ void k_one(const u32* a, const u32* b, u32* o) {
  Fe<P> x, y; for (int i = 0; i < 9; i++) { x.l[i] = a[threadIdx.x * 9 + i]; y.l[i] = b[threadIdx.x * 9 + i]; }
  Fe<P> r = fe_mul<P>(x, y);
  for (int i = 0; i < 9; i++) o[threadIdx.x * 9 + i] = r.l[i];
}
