// mzk_inv_wave.h -- ONE field inversion on ONE wave (device only): the safegcd of mzk_field.h (Bernstein-Yang divsteps, 30 per
// batch) with its multi-limb work spread over the lanes of the wave.
//
// Every MSM, fold and small commit ends in one Fq inversion (XYZZ -> affine), a ~9000-instruction dependent chain on a single
// lane: 35 us of the ~46 us conversion (profiles/r03l_*).  Half of those instructions are the per-batch updates of f, g, d, e
// (9 limbs each, limb-serial carries).  Here the four numbers live in the four 16-lane rows of the wave, limb i in lane i of
// the row, so the whole update of a batch is three v_mad_i64_i32 per lane (the partner number arrives by v_permlane16_swap)
// and ONE re-cut of the 64-bit lane values at 30-bit boundaries (low piece one lane down, top piece one lane up, DPP),
// followed by an exact carry resolution from generate / propagate ballots (the sign of d, e and the zero test of g need
// canonical limbs; lanes 9..15 of a row hold zeros, so no carry crosses a row); the 30 divsteps of a batch run on wave-uniform
// values (f_0, g_0 by v_readlane), i.e. on the scalar unit.
// Same algorithm, same batches, same result as fe_inv_safegcd (checked by mzk_selftest_inv_wave).
//
// Call with all 64 lanes of a wave active and the SAME argument in every lane; every lane returns the result.
#pragma once
#include "mzk_field.h"

namespace mzk {
namespace invw {

template <int CTRL> __device__ __forceinline__ u32 dpp32(u32 v) {
  u32 r = (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ u32 up1(u32 v) { return dpp32<0x111>(v); }     // row_shr:1  lane j <- lane j - 1
__device__ __forceinline__ u32 down1(u32 v) { return dpp32<0x101>(v); }   // row_shl:1  lane j <- lane j + 1

constexpr u32 M30 = 0x3fffffffu;

// lane i holds t_i (signed 64-bit, |t_i| < 2^62 - 2^32; |t_8| < 2^61); the number is sum t_i 2^(30 i), divisible by 2^30.
// Returns limb j of (number / 2^30) in lane j: limbs 0..7 in [0, 2^30), limb 8 signed (the top limb carries the sign), lanes >= 9
// zero -- the canonical form the serial sg_update_* produce.
// Negative lane values would need borrows as well as carries.  A telescoping bias avoids them without changing the number:
// lane i adds 2^62 and lane i + 1 takes 2^32 = 2^62 / 2^30 away (lane 0 only adds, lane 8 only subtracts), so lanes 0..7 are
// non-negative and every 30-bit piece is.
__device__ __forceinline__ int64_t recut_bias(int j) {
  const int64_t B = (int64_t)1 << 62, b = (int64_t)1 << 32;
  return (j == 0) ? B : (j < 8) ? (B - b) : (j == 8) ? -b : 0;
}
__device__ __forceinline__ i32 recut_div30(int64_t t_in, int j, int64_t bias) {
  const int64_t t = t_in + bias;
  const u32 lo = (u32)t & M30;
  const u32 mid = (u32)(t >> 30) & M30;
  const u32 top = (u32)((uint64_t)t >> 60);                    // lanes 0..7: 0 .. 7
  const i32 full = (i32)(t >> 30);                             // lane 8 only: its value above bit 30 fits (|t_8| < 2^50)
  const u32 r = down1(lo) + ((j == 8) ? (u32)full : mid) + up1(top);      // lanes 0..7: 0 .. 2^31 + 5; lane 8: signed
  // one parallel carry step, then the remaining 0/1 carries from generate / propagate masks (as rowop::exact)
  const u32 c = (j < 8) ? (r >> 30) : 0u;
  u32 y = ((j < 8) ? (r & M30) : r) + up1(c);
  const bool g = (j < 8) && (y > M30), p = (j < 8) && (y == M30);
  const u64 G = __builtin_amdgcn_ballot_w64(g), Pm = __builtin_amdgcn_ballot_w64(p);
  const u64 A = G | Pm;
  const u64 cin = (A + G) ^ A ^ G;
  y += (u32)((cin >> (threadIdx.x & 63)) & 1ull);
  if (j < 8) y &= M30;
  return (j < 9) ? (i32)y : 0;
}

template <class P> __device__ __forceinline__ Fe<P> inv(const Fe<P>& a) {
  constexpr int NW = P::NW;
  constexpr int NL = (32 * NW + 29) / 30;
  static_assert(NL <= 9, "one row of 16 lanes holds the limbs");
  const int lane = (int)(threadIdx.x & 63), j = lane & 15, row = lane >> 4;
  u32 xw[NW], pw[NW];
  {
    Fe<P> c = fe_reduce<P>(a);
    if (fe_is_zero_canon<P>(c)) return fe_zero<P>();
    fe_pack<P>(c, xw);
  }
#pragma unroll
  for (int i = 0; i < NW; i++) pw[i] = P::PW[i];
  Sgn30<NL> mod, x0;
  sg_from_words<NW, NL>(pw, &mod);
  sg_from_words<NW, NL>(xw, &x0);
  // the four numbers of the algorithm in the four rows of the wave, limb i in lane i of the row:
  // row 0 = f (= p), row 1 = g (= x), row 2 = d (= 0), row 3 = e (= 1)
  i32 modl = 0, xl = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) { modl = (j == i) ? mod.v[i] : modl; xl = (j == i) ? x0.v[i] : xl; }
  i32 x = (row == 0) ? modl : (row == 1) ? xl : (row == 3 && j == 0) ? 1 : 0;
  const bool odd = (row & 1) != 0, de = (row & 2) != 0;
  u32 pinv = (u32)mod.v[0];
#pragma unroll
  for (int it = 0; it < 4; it++) pinv *= 2 - (u32)mod.v[0] * pinv;
  pinv &= M30;
  i32 eta = -1;
  const int64_t bias = recut_bias(j);
  for (int batch = 0; batch < 32; batch++) {
    // 30 divsteps on the low limbs: wave-uniform, scalar unit
    DivMat t;
    const u32 f0 = (u32)__builtin_amdgcn_readlane(x, 0), g0 = (u32)__builtin_amdgcn_readlane(x, 16);
    eta = sg_divsteps_30(eta, f0, g0, &t);
    const i32 d0 = __builtin_amdgcn_readlane(x, 32), e0 = __builtin_amdgcn_readlane(x, 48);
    const i32 sd = __builtin_amdgcn_readlane(x, 32 + NL - 1) >> 31, se = __builtin_amdgcn_readlane(x, 48 + NL - 1) >> 31;
    i32 md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    md -= (i32)((pinv * (u32)((int64_t)t.u * d0 + (int64_t)t.v * e0) + (u32)md) & M30);
    me -= (i32)((pinv * (u32)((int64_t)t.q * d0 + (int64_t)t.r * e0) + (u32)me) & M30);
    // rows 0, 1: t (f, g); rows 2, 3: t (d, e) + (md, me) p -- each lane its limb of its row's number (exact multiples of
    // 2^30), then ONE re-cut divides all four by 2^30.  The partner number comes by v_permlane16_swap:
    // [f g d e] -> [f f d d], [g g e e].
    const auto sw = __builtin_amdgcn_permlane16_swap((u32)x, (u32)x, false, false);
    const i32 first = (i32)sw[0], second = (i32)sw[1];
    const i32 ca = odd ? t.q : t.u, cb = odd ? t.r : t.v, cm = de ? (odd ? me : md) : 0;
    const int64_t tt = (int64_t)ca * first + (int64_t)cb * second + (int64_t)cm * modl;
    x = recut_div30(tt, j, bias);
    if ((__builtin_amdgcn_ballot_w64(x != 0) & 0x00000000ffff0000ull) == 0) break;       // g == 0
  }
  // f = +-1, d = +-x^-1 in (-2p, p): the few remaining steps on wave-uniform scalars, as fe_inv_safegcd
  Sgn30<NL> dd;
#pragma unroll
  for (int i = 0; i < NL; i++) dd.v[i] = __builtin_amdgcn_readlane(x, 32 + i);
  sg_normalize<NL>(&dd, __builtin_amdgcn_readlane(x, NL - 1), &mod);
  u32 rw[NW];
  sg_to_words<NW, NL>(&dd, rw);
  Fe<P> r3;
#pragma unroll
  for (int i = 0; i < P::L; i++) r3.l[i] = P::R3[i];
  return fe_mul<P>(fe_unpack<P>(rw), r3);
}

}  // namespace invw
}  // namespace mzk
