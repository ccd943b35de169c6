// mzk_msm_tail.hip -- the single-wave tail of the MSM: Horner over the bucket sets (4-lane cooperative
// doubling), the final Fq inversion, and the multi-GPU fold of XYZZ partials.  A separate translation
// unit only to keep compile times of mzk_msm.hip down.
#include "mzk_common.h"
#include "mzk_ec.h"
#include "mzk_coop.h"

namespace mzk {


// ---- global loads of packed 256-bit values -----------------------------------------------------------
__device__ __forceinline__ void load_words8(const u32* __restrict__ g, u32* w) {
  const uint4* p4 = reinterpret_cast<const uint4*>(g);
  uint4 a = p4[0], b = p4[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store_words8(u32* __restrict__ g, const u32* w) {
  uint4* p4 = reinterpret_cast<uint4*>(g);
  p4[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p4[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ Xyzz xyzz_gload(const u32* __restrict__ g, size_t idx) {
  u32 w[32];
#pragma unroll
  for (int q = 0; q < 4; q++) load_words8(g + idx * 32 + 8 * q, w + 8 * q);
  return xyzz_load(w);
}
__device__ __forceinline__ void xyzz_gstore(u32* __restrict__ g, size_t idx, const Xyzz& p) {
  u32 w[32];
  xyzz_store(p, w);
#pragma unroll
  for (int q = 0; q < 4; q++) store_words8(g + idx * 32 + 8 * q, w + 8 * q);
}


// ---- 6. window combine ----------------------------------------------------------------------------------
// total = sum_w 2^(c w) R_w  (Horner, c doublings per window; a single bucket set skips it), then affine
// (one Fq inversion) or the XYZZ partial record.
//
// The 240 doublings are inherently serial, so the lever is the latency of ONE doubling: its 9 field
// products form only 3 dependency levels; a DPP quad holds the same point and splits each level
// (xyzz_dbl_quad / xyzz_add_quad, mzk_coop.h).  One quad does the work.
__global__ __launch_bounds__(4) void k_window_combine(const u32* __restrict__ wsum, int nwin, int c, int out_xyzz, u32* __restrict__ out) {
  const int lane = threadIdx.x & 3;
  Xyzz tot = xyzz_gload_quad(wsum, nwin - 1, lane);
  for (int win = nwin - 2; win >= 0; win--) {
    for (int d = 0; d < c; d++) tot = xyzz_dbl_quad(tot, lane);
    tot = xyzz_add_quad(tot, xyzz_gload_quad(wsum, win, lane), lane);
  }
  if (threadIdx.x != 0) return;
  if (out_xyzz) {
    u32 wds[32];
    xyzz_store(tot, wds);
    for (int i = 0; i < 32; i++) out[i] = wds[i];
  } else {
    u32 wds[16];
    Affine af;
    if (xyzz_to_affine<true>(tot, &af)) affine_store_plain(af, wds);
    else for (int i = 0; i < 16; i++) wds[i] = 0;
    for (int i = 0; i < 16; i++) out[i] = wds[i];
  }
}
// fold `count` XYZZ partials (multi-GPU all-gather result) into one affine point; one DPP quad does the chain
__global__ __launch_bounds__(4) void k_fold_partials(const u32* __restrict__ partials, int count, u32* __restrict__ out) {
  const int lane = threadIdx.x & 3;
  Xyzz tot = xyzz_inf();
  for (int i = 0; i < count; i++) tot = xyzz_add_quad(tot, xyzz_gload_quad(partials, i, lane), lane);
  if (threadIdx.x != 0) return;
  u32 wds[16];
  Affine af;
  if (xyzz_to_affine<true>(tot, &af)) affine_store_plain(af, wds);
  else for (int i = 0; i < 16; i++) wds[i] = 0;
  for (int i = 0; i < 16; i++) out[i] = wds[i];
}


int launch_window_combine(const u32* wsum, int nwin, int c, int out_xyzz, u32* out, hipStream_t s) {
  hipLaunchKernelGGL(k_window_combine, dim3(1), dim3(4), 0, s, wsum, nwin, c, out_xyzz, out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int launch_fold_partials(const u32* partials, int count, u32* out, hipStream_t s) {
  hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(4), 0, s, partials, count, out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

}  // namespace mzk
