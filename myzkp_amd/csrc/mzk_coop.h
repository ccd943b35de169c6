// mzk_coop.h -- quad-cooperative XYZZ group operations (device only).
//
// The bucket reduction and every MSM tail are chains of DEPENDENT group operations with little parallelism
// left (15 halving levels, Horner, tree sums): what matters is the latency of one operation, and a general
// addition is 14 dependent-issue field products on one lane.  Its products form only four dependency levels,
// so four adjacent lanes (a DPP quad) hold the same operands, each computes a different product per level
// (same instruction stream, lane-selected operands) and quad_perm DPP moves broadcast the results:
// 4 product latencies per addition instead of 14, 3 per doubling instead of 9.
//
// All four lanes of a quad must be active (a DPP read of a disabled lane returns 0), so callers assign work
// per quad and keep control flow quad-uniform; the functions themselves are straight-line (exceptional cases
// are resolved by selects; the P + P case, which needs the doubling formula, is taken by the whole wave when
// any quad hits it).
#pragma once
#include "mzk_ec.h"

namespace mzk {

template <int K> __device__ __forceinline__ u32 quad_bcast_u32(u32 v) {
  u32 r = (u32)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, true);   // quad_perm [K,K,K,K]
  // Keep it a plain v_mov_b32_dpp: when the DPP-combine pass folded the broadcast into a consuming v_sub_u32
  // (the lazy subtraction a + k p - b, b broadcast) the result was wrong on gfx950 with ROCm 7.2
  // (tools/microbench/dpp_combine_miscompile_repro.hip: Y3 differed per lane).  The empty asm makes the value opaque to that pass.
  asm volatile("" : "+v"(r));
  return r;
}
template <int K> __device__ __forceinline__ Fq quad_bcast(const Fq& v) {
  Fq r;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) r.l[i] = quad_bcast_u32<K>(v.l[i]);
  return r;
}
__device__ __forceinline__ Fq quad_sel(int lane, const Fq& a0, const Fq& a1, const Fq& a2, const Fq& a3) {
  Fq r;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) {
    // selects on VALUES: a conditional expression over lvalues is an lvalue, i.e. a select of addresses followed
    // by a load, which pins the operands in scratch memory
    const u32 v0 = a0.l[i], v1 = a1.l[i], v2 = a2.l[i], v3 = a3.l[i];
    const u32 lo = (lane & 1) ? v1 : v0, hi = (lane & 1) ? v3 : v2;
    r.l[i] = (lane & 2) ? hi : lo;
  }
  return r;
}
__device__ __forceinline__ Xyzz xyzz_select(bool c, const Xyzz& a, const Xyzz& b) {   // c ? a : b, limbwise
  Xyzz r;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) {
    const u32 ax = a.X.l[i], ay = a.Y.l[i], az = a.ZZ.l[i], aw = a.ZZZ.l[i];
    const u32 bx = b.X.l[i], by = b.Y.l[i], bz = b.ZZ.l[i], bw = b.ZZZ.l[i];
    r.X.l[i] = c ? ax : bx;
    r.Y.l[i] = c ? ay : by;
    r.ZZ.l[i] = c ? az : bz;
    r.ZZZ.l[i] = c ? aw : bw;
  }
  return r;
}

// Quad-cooperative load / store of XYZZ record `idx`: lane k moves coordinate k (32 bytes) and the quad
// exchanges them, instead of every lane moving all 128 bytes.
__device__ __forceinline__ Xyzz xyzz_gload_quad(const u32* __restrict__ g, size_t idx, int lane) {
  u32 w[8];
  const uint4* p4 = reinterpret_cast<const uint4*>(g + idx * 32 + 8 * lane);
  uint4 a = p4[0], b = p4[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  const Fq mine = fe_unpack<FqParams>(w);
  Xyzz p;
  p.X = quad_bcast<0>(mine);
  p.Y = quad_bcast<1>(mine);
  p.ZZ = quad_bcast<2>(mine);
  p.ZZZ = quad_bcast<3>(mine);
  return p;
}
__device__ __forceinline__ void xyzz_gstore_quad(u32* __restrict__ g, size_t idx, const Xyzz& p, int lane) {
  // same packing as xyzz_store: canonical words per coordinate
  u32 w[8];
  fe_pack<FqParams>(fe_reduce<FqParams>(quad_sel(lane, p.X, p.Y, p.ZZ, p.ZZZ)), w);
  if (xyzz_is_inf(p)) {                  // canonical all-zero record for infinity, as xyzz_store writes it
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = 0;
  }
  uint4* p4 = reinterpret_cast<uint4*>(g + idx * 32 + 8 * lane);
  p4[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p4[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// p replicated in the quad -> 2p replicated (dbl-2008-s-1; same schedule as the wave-wide variant in
// mzk_msm_tail.hip).  Infinity in -> infinity out via the final select.
__device__ __forceinline__ Xyzz xyzz_dbl_quad(const Xyzz& p, int lane) {
  typedef FqParams P;
  const bool was_inf = xyzz_is_inf(p);
  const Fq U = fe_dbl<P>(p.Y);
  Fq r = fe_sqr<P>(quad_sel(lane, U, p.X, U, p.X));                       // V = U^2 | X^2
  const Fq V = quad_bcast<0>(r), X2 = quad_bcast<1>(r);
  const Fq M = fe_carry<P>(fe_add<P>(fe_dbl<P>(X2), X2));                 // 3 X^2
  r = fe_mul<P>(quad_sel(lane, U, p.X, M, M), quad_sel(lane, V, V, M, M));   // W = U V | S = X V | M^2
  const Fq W = quad_bcast<0>(r), S = quad_bcast<1>(r), MM = quad_bcast<2>(r);
  const Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(MM, S), S));
  const Fq Vd = fe_carry<P>(fe_sub<P, 8>(S, X3));
  r = fe_mul<P>(quad_sel(lane, M, W, V, W), quad_sel(lane, Vd, p.Y, p.ZZ, p.ZZZ));   // A | B | ZZ3 | ZZZ3
  const Fq A = quad_bcast<0>(r), B = quad_bcast<1>(r);
  Xyzz o;
  o.X = X3;
  o.Y = fe_weak_reduce<P>(fe_sub<P, 4>(A, B));
  o.ZZ = quad_bcast<2>(r);
  o.ZZZ = quad_bcast<3>(r);
  return xyzz_select(was_inf, xyzz_inf(), o);
}

// a, b replicated in the quad -> a + b replicated.  add-2008-s in four product levels:
//   1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
//   2: PP = P^2    | RR = R^2    | ZZ1 ZZ2      | ZZZ1 ZZZ2            (P = U2 - U1, R = S2 - S1)
//   3: PPP = P PP  | Q = U1 PP   | ZZ3 = (ZZ1 ZZ2) PP
//   4: R (Q - X3)  | S1 PPP      | ZZZ3 = (ZZZ1 ZZZ2) PPP              X3 = RR - PPP - 2Q, Y3 = first - second
// Exception-complete like xyzz_add (curve.rs:104-115): inf + b = b, a + inf = a, a + (-a) = inf, a + a = 2a.
__device__ __forceinline__ Xyzz xyzz_add_quad(const Xyzz& a, const Xyzz& b, int lane) {
  typedef FqParams P;
  const bool a_inf = xyzz_is_inf(a), b_inf = xyzz_is_inf(b);
  Fq r = fe_mul<P>(quad_sel(lane, a.X, b.X, a.Y, b.Y), quad_sel(lane, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ));
  const Fq U1 = quad_bcast<0>(r), U2 = quad_bcast<1>(r), S1 = quad_bcast<2>(r), S2 = quad_bcast<3>(r);
  const Fq Pd = fe_carry<P>(fe_sub<P, 4>(U2, U1));                          // < 5.04
  const Fq Rd = fe_carry<P>(fe_sub<P, 4>(S2, S1));
  const bool p_zero = fe_is_zero_mod<P, 6>(Pd), r_zero = fe_is_zero_mod<P, 6>(Rd);
  r = fe_mul<P>(quad_sel(lane, Pd, Rd, a.ZZ, a.ZZZ), quad_sel(lane, Pd, Rd, b.ZZ, b.ZZZ));
  const Fq PP = quad_bcast<0>(r), RR = quad_bcast<1>(r), ZZp = quad_bcast<2>(r), ZZZp = quad_bcast<3>(r);
  r = fe_mul<P>(quad_sel(lane, Pd, U1, ZZp, ZZp), PP);                      // PPP | Q | ZZ3 | (ZZ3 again)
  const Fq PPP = quad_bcast<0>(r), Q = quad_bcast<1>(r), ZZ3 = quad_bcast<2>(r);
  const Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(fe_sub<P, 4>(RR, PPP), Q), Q));
  const Fq Vd = fe_carry<P>(fe_sub<P, 8>(Q, X3));                           // < 9.02
  r = fe_mul<P>(quad_sel(lane, Rd, S1, ZZZp, ZZZp), quad_sel(lane, Vd, PPP, PPP, PPP));   // A | B | ZZZ3
  const Fq A = quad_bcast<0>(r), B = quad_bcast<1>(r);
  Xyzz o;
  o.X = X3;
  o.Y = fe_weak_reduce<P>(fe_sub<P, 4>(A, B));
  o.ZZ = ZZ3;
  o.ZZZ = quad_bcast<2>(r);
  const bool generic = !a_inf && !b_inf;
  const bool need_dbl = generic && p_zero && r_zero;
  if (__builtin_amdgcn_ballot_w64(need_dbl) != 0) {      // wave-uniform: some quad adds a point to itself
    const Xyzz d = xyzz_dbl(a);                           // plain (non-cooperative) doubling, no cross-lane traffic
    o = xyzz_select(need_dbl, d, o);
  }
  if (__builtin_amdgcn_ballot_w64(a_inf || b_inf || p_zero) != 0) {       // wave-uniform: the three select chains only where some quad needs one
    o = xyzz_select(generic && p_zero && !r_zero, xyzz_inf(), o);
    o = xyzz_select(b_inf, a, o);
    o = xyzz_select(a_inf, b, o);
  }
  return o;
}

}  // namespace mzk
