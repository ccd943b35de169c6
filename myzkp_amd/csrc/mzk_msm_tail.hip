// mzk_msm_tail.hip -- the single-LANE tail of the MSM: Horner over the bucket sets, the final Fq inversion,
// and the multi-GPU fold of XYZZ partials.
// Built with MZK_COMPACT_CODE: fe_mul / fe_sqr / xyzz_add / xyzz_dbl are real functions here (see
// mzk_field.h), which shrinks these kernels ~10x; one lane cannot hide instruction-cache misses (window
// Horner 2.8 -> 2.1 ms).  The multi-wave k_reduce_tail stays inlined in mzk_msm.hip: there the calls cost
// more than the fetches (0.32 -> 0.41 ms measured).
#define MZK_COMPACT_CODE 1
#include "mzk_common.h"
#include "mzk_ec.h"

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
__global__ void k_window_combine(const u32* __restrict__ wsum, int nwin, int c, int out_xyzz, u32* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Xyzz tot = xyzz_gload(wsum, nwin - 1);
  for (int win = nwin - 2; win >= 0; win--) {
    for (int d = 0; d < c; d++) tot = xyzz_dbl(tot);
    tot = xyzz_add(tot, xyzz_gload(wsum, win));
  }
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
// fold `count` XYZZ partials (multi-GPU all-gather result) into one affine point
__global__ void k_fold_partials(const u32* __restrict__ partials, int count, u32* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Xyzz tot = xyzz_inf();
  for (int i = 0; i < count; i++) tot = xyzz_add(tot, xyzz_gload(partials, i));
  u32 wds[16];
  Affine af;
  if (xyzz_to_affine<true>(tot, &af)) affine_store_plain(af, wds);
  else for (int i = 0; i < 16; i++) wds[i] = 0;
  for (int i = 0; i < 16; i++) out[i] = wds[i];
}


int launch_window_combine(const u32* wsum, int nwin, int c, int out_xyzz, u32* out, hipStream_t s) {
  hipLaunchKernelGGL(k_window_combine, dim3(1), dim3(64), 0, s, wsum, nwin, c, out_xyzz, out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int launch_fold_partials(const u32* partials, int count, u32* out, hipStream_t s) {
  hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(64), 0, s, partials, count, out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

}  // namespace mzk
