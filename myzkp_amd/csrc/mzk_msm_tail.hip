// mzk_msm_tail.hip -- the single-wave tail of the MSM: Horner over the bucket sets (4-lane cooperative
// doubling), the final Fq inversion, and the multi-GPU fold of XYZZ partials.  A separate translation
// unit only to keep compile times of mzk_msm.hip down.
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
//
// The 240 doublings are inherently serial, so the lever is the latency of ONE doubling: its 9 field
// products form only 3 dependency levels (V, X^2 | W, S, M^2 | M(S-X3), W Y, V ZZ, W ZZZ).  Four lanes
// hold the same point; at each level every lane computes a different product (same instruction stream,
// lane-selected operands) and the results are broadcast back with v_readlane.  3 product latencies per
// doubling instead of 9.
__device__ __forceinline__ Fq bcast_lane(const Fq& v, int src) {
  Fq r;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) r.l[i] = (u32)__builtin_amdgcn_readlane((int)v.l[i], src);
  return r;
}
__device__ __forceinline__ Fq sel4(int lane, const Fq& a0, const Fq& a1, const Fq& a2, const Fq& a3) {
  Fq r;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) r.l[i] = (lane == 0) ? a0.l[i] : (lane == 1) ? a1.l[i] : (lane == 2) ? a2.l[i] : a3.l[i];
  return r;
}
// p is replicated in lanes 0..3 (lane = index within the group of four); result replicated again.
// Straight-line on purpose: the cross-lane reads must not sit behind a branch the compiler cannot prove
// uniform (an early `return` for infinity miscompiled when this was inlined into the Horner loop).
// Infinity needs no special case: all-zero coordinates give ZZ3 = V * 0 = 0 exactly, and the final
// select restores the canonical all-zero encoding.
__device__ __forceinline__ void xyzz_dbl_coop4(Xyzz& p, int lane) {
  typedef FqParams P;
  const bool was_inf = xyzz_is_inf(p);
  const Fq U = fe_dbl<P>(p.Y);                                  // < 5, limbs < 2^30
  Fq r = fe_sqr<P>(sel4(lane, U, p.X, U, p.X));                 // lane 0: V = U^2, lane 1: X^2
  const Fq V = bcast_lane(r, 0), X2 = bcast_lane(r, 1);
  const Fq M = fe_carry<P>(fe_add<P>(fe_dbl<P>(X2), X2));       // 3 X^2 < 3.12, N
  r = fe_mul<P>(sel4(lane, U, p.X, M, M), sel4(lane, V, V, M, M));   // W = U V | S = X V | M^2
  const Fq W = bcast_lane(r, 0), S = bcast_lane(r, 1), MM = bcast_lane(r, 2);
  const Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(MM, S), S));
  const Fq Vd = fe_carry<P>(fe_sub<P, 8>(S, X3));               // < 9.02
  r = fe_mul<P>(sel4(lane, M, W, V, W), sel4(lane, Vd, p.Y, p.ZZ, p.ZZZ));   // A | B | ZZ3 | ZZZ3
  const Fq A = bcast_lane(r, 0), B = bcast_lane(r, 1);
  p.ZZ = bcast_lane(r, 2);
  p.ZZZ = bcast_lane(r, 3);
  p.X = X3;
  p.Y = fe_weak_reduce<P>(fe_sub<P, 4>(A, B));                  // A < 1.17, B < 1.02
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    p.X.l[i] = was_inf ? 0u : p.X.l[i];
    p.Y.l[i] = was_inf ? 0u : p.Y.l[i];
    p.ZZ.l[i] = was_inf ? 0u : p.ZZ.l[i];
    p.ZZZ.l[i] = was_inf ? 0u : p.ZZZ.l[i];
  }
}
__global__ __launch_bounds__(64) void k_window_combine(const u32* __restrict__ wsum, int nwin, int c, int out_xyzz, u32* __restrict__ out) {
  const int lane = threadIdx.x & 3;     // every group of four lanes replicates the same computation
  Xyzz tot = xyzz_gload(wsum, nwin - 1);
  for (int win = nwin - 2; win >= 0; win--) {
    for (int d = 0; d < c; d++) xyzz_dbl_coop4(tot, lane);
    tot = xyzz_add(tot, xyzz_gload(wsum, win));
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
