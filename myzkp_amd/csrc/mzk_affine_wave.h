// mzk_affine_wave.h -- the last step of every MSM: one XYZZ record -> the canonical affine point of the ABI, on ONE WAVE
// (the inversion of ZZ ZZZ spread over its lanes: mzk_inv_wave.h).  Shared by the tails of mzk_msm_row.hip and the finish
// kernel of the direct-table batches in mzk_msm.hip.
#pragma once
#include "mzk_ec.h"
#include "mzk_inv_wave.h"

namespace mzk {

// XYZZ record (packed, in LDS or global memory) -> canonical affine point at the ABI (plain words, all-zero = infinity), by ONE
// WAVE: the inversion of ZZ ZZZ is spread over the lanes (mzk_inv_wave.h), the six products around it run redundantly in every
// lane.  All 64 lanes must be active; lane 0 stores.
__device__ __forceinline__ void wave_store_affine(const u32* rec, u32* __restrict__ out) {
  typedef FqParams P;
  u32 w[32];
#pragma unroll
  for (int i = 0; i < 32; i++) w[i] = rec[i];
  const Xyzz p = xyzz_load(w);
  u32 wds[16];
  if (xyzz_is_inf(p)) {
#pragma unroll
    for (int i = 0; i < 16; i++) wds[i] = 0;
  } else {
    const Fq di = invw::inv<P>(fe_mul<P>(p.ZZ, p.ZZZ));          // 1 / (ZZ ZZZ)
    const Fq izz = fe_mul<P>(di, p.ZZZ), izzz = fe_mul<P>(di, p.ZZ);
    Affine af;
    af.x = fe_reduce<P>(fe_mul<P>(p.X, izz));
    af.y = fe_reduce<P>(fe_mul<P>(p.Y, izzz));
    affine_store_plain(af, wds);
  }
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 16; i++) out[i] = wds[i];
}

}  // namespace mzk
