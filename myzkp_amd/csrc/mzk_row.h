// mzk_row.h -- ROW-cooperative Fq arithmetic and XYZZ group operations (device only): ONE point operation per wave.
//
// The tails of every MSM (bucket-reduction halving steps, the 2^j weights, tree sums, the window Horner, the multi-GPU fold)
// are chains of DEPENDENT group operations; what counts is the latency of one.  The quad form (mzk_coop.h) still runs every
// field product on one lane: 214 dependent instructions, four product levels per addition, and as many instructions again
// in limb-wise additions, selects and broadcasts (~2000 per addition).  Here a field element is spread over the 16 lanes of
// a DPP row -- lane j holds 29-bit limb j, lanes 9..15 zero ("distributed", D) -- so that
//   * additions, subtractions, selects are ONE instruction for all nine limbs;
//   * a product is column-parallel: lane j accumulates column j of a*b with nine v_mad_u64_u32 (operand a replicated in the
//     row by nine row_newbcast moves, operand b shifted by row_shr), and the Montgomery reduction needs no limb-serial
//     loop either: the low half is normalised by splitting every 64-bit column into three 29-bit pieces and shifting them one
//     and two lanes up, m = low * (-p^-1 mod R) is a second column product, m * p a third, and the carry of the low half
//     into the result is read off ONE limb (tools/row_product_model.py proves the bounds): ~100 instructions deep;
//   * the four rows of a wave compute the (up to) four independent products of a level of the addition formula at once,
//     and v_permlane16/32_swap hands every row all four results.
// An addition is four such levels (~700 instructions, ~1.3 us on an otherwise idle GPU against ~3.6 us), a doubling three.
//
// Exceptional cases exactly as the reference distinguishes them (curve.rs:104-115): infinity operands are wave-uniform
// branches; P == +-Q is detected by a one-limb filter (x = k p  =>  x_0 p_0^-1 mod 2^29 = k <= KMAX) and then resolved by
// the plain exception-complete formulas of mzk_ec.h on scalars read from the row (rare: two equal points met in a sum).
//
// Records in memory are the packed XYZZ records every other kernel uses (4 x 8 words, normalised, value < 2.5 p, all-zero =
// infinity), so row kernels and quad / plain kernels can follow one another on the same buffers.
// Value bounds (units of p, rho = p / R = 2^-7.4; a product of values (a, b) is < a b rho + 2.01): stored coordinates < 2.5;
// the bounds of every intermediate are written beside the formulas below.
#pragma once
#include "mzk_ec.h"

namespace mzk {
namespace rowop {

typedef FqParams P;
typedef FqRowParams RP;

template <int CTRL> __device__ __forceinline__ u32 dpp_mov(u32 v) {
  u32 r = (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);      // invalid source lanes read as 0
  asm volatile("" : "+v"(r));       // keep it a plain v_mov_b32_dpp (see quad_bcast_u32 in mzk_coop.h: DPP-combine miscompile)
  return r;
}
template <int N> __device__ __forceinline__ u32 row_shr(u32 v) { if constexpr (N == 0) return v; else return dpp_mov<0x110 + N>(v); }   // lane j <- lane j - N
template <int N> __device__ __forceinline__ u32 row_shl(u32 v) { if constexpr (N == 0) return v; else return dpp_mov<0x100 + N>(v); }   // lane j <- lane j + N
template <int N> __device__ __forceinline__ u32 row_ror(u32 v) { return dpp_mov<0x120 + N>(v); }    // lane j <- lane (j - N) mod 16
template <int N> __device__ __forceinline__ u32 row_bcast(u32 v) { return dpp_mov<0x150 + N>(v); }  // row_newbcast: every lane <- lane N of its row

// per-lane constants of a wave that runs row operations
struct Lane {
  int j, row;
  u32 lt8, lt9;           // all-ones below limb 8 / 9
  u32 pj;                 // unused limbs are zero
  u32 one;                // R mod p, limb j
  int wk, ws;             // unpack: limb j starts at bit ws of word wk
  u32 hi_ok;              // all-ones where word wk + 1 exists
};
__device__ __forceinline__ Lane lane_init() {
  Lane ln;
  const int lane = (int)(threadIdx.x & 63);
  ln.j = lane & 15;
  ln.row = lane >> 4;
  ln.lt8 = ln.j < 8 ? ~0u : 0u;
  ln.lt9 = ln.j < 9 ? ~0u : 0u;
  ln.pj = RP::P16[ln.j];
  ln.one = RP::ONE16[ln.j];
  const int jj = ln.j < 8 ? ln.j : 8;
  ln.wk = (29 * jj) >> 5;
  ln.ws = (29 * jj) & 31;
  ln.hi_ok = ln.j < 8 ? ~0u : 0u;
  return ln;
}
template <int K> __device__ __forceinline__ u32 kps(const Lane& ln) { return RP::KPS[K][ln.j]; }

__device__ __forceinline__ u32 sel4(const Lane& ln, u32 v0, u32 v1, u32 v2, u32 v3) {
  const u32 lo = (ln.row & 1) ? v1 : v0, hi = (ln.row & 1) ? v3 : v2;
  return (ln.row & 2) ? hi : lo;
}
// every row's value of `v` (row r holds x_r) -> g_r = x_r in ALL rows: v_permlane16_swap + 2 v_permlane32_swap
__device__ __forceinline__ void gather4(u32 v, u32& g0, u32& g1, u32& g2, u32& g3) {
  const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);          // [x0 x0 x2 x2], [x1 x1 x3 x3]
  const auto t0 = __builtin_amdgcn_permlane32_swap(s[0], s[0], false, false);   // [x0 x0 x0 x0], [x2 x2 x2 x2]
  const auto t1 = __builtin_amdgcn_permlane32_swap(s[1], s[1], false, false);
  g0 = t0[0]; g2 = t0[1]; g1 = t1[0]; g3 = t1[1];
}

struct Rep { u32 l[9]; };       // an element replicated in its row: every lane holds all nine limbs
__device__ __forceinline__ Rep rep(u32 d) {
  Rep r;
  r.l[0] = row_bcast<0>(d); r.l[1] = row_bcast<1>(d); r.l[2] = row_bcast<2>(d); r.l[3] = row_bcast<3>(d); r.l[4] = row_bcast<4>(d);
  r.l[5] = row_bcast<5>(d); r.l[6] = row_bcast<6>(d); r.l[7] = row_bcast<7>(d); r.l[8] = row_bcast<8>(d);
  return r;
}

__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) { return c + (u64)a * b; }
// Three independent chains of three multiply-adds (a single wave waits ~8 cycles for every DEPENDENT v_mad_u64_u32; the row
// operations run alone on their SIMD), summed at the end -- same column sum.
__device__ __forceinline__ u64 col_ab(const Rep& a, u32 b, u64 acc) {      // lane j: acc += sum_i a_i b_(j - i)
  u64 s0 = mad64(a.l[0], b, acc), s1 = mad64(a.l[3], row_shr<3>(b), 0), s2 = mad64(a.l[6], row_shr<6>(b), 0);
  s0 = mad64(a.l[1], row_shr<1>(b), s0); s1 = mad64(a.l[4], row_shr<4>(b), s1); s2 = mad64(a.l[7], row_shr<7>(b), s2);
  s0 = mad64(a.l[2], row_shr<2>(b), s0); s1 = mad64(a.l[5], row_shr<5>(b), s1); s2 = mad64(a.l[8], row_shr<8>(b), s2);
  return s0 + s1 + s2;
}
// lane j: sum_i C[i] v_(j - i) for a compile-time constant C (N' or p)
template <const u32 (&C)[9]> __device__ __forceinline__ u64 col_const(u32 v, u64 acc) {
  u64 s0 = mad64(C[0], v, acc), s1 = mad64(C[3], row_shr<3>(v), 0), s2 = mad64(C[6], row_shr<6>(v), 0);
  s0 = mad64(C[1], row_shr<1>(v), s0); s1 = mad64(C[4], row_shr<4>(v), s1); s2 = mad64(C[7], row_shr<7>(v), s2);
  s0 = mad64(C[2], row_shr<2>(v), s0); s1 = mad64(C[5], row_shr<5>(v), s1); s2 = mad64(C[8], row_shr<8>(v), s2);
  return s0 + s1 + s2;
}
// n_j = lo(acc_j) + mid(acc_(j-1)) + hi(acc_(j-2)): the column sums re-cut at 29-bit boundaries without a ripple
__device__ __forceinline__ u32 recut(u64 acc) {
  const u32 lo = (u32)acc & MASK29, mid = (u32)(acc >> W29) & MASK29, hi = (u32)(acc >> (2 * W29));
  return lo + row_shr<1>(mid) + row_shr<2>(hi);
}
// Montgomery product(s) of row-distributed operands: (a b [+ a2 b2]) / R mod p, distributed, limbs < 2^30 + 64 ("lazy"), value
// < (a b + a2 b2) rho + 2.01 p.  Operand limbs: a, b < 2^30 + 64 for one product; all four < 2^29.6 for the fused pair except
// that ONE of them may be a lazy product output (columns: tools/row_product_model.py).  b, b2 must be zero in lanes 9..15.
template <bool TWO> __device__ __forceinline__ u32 mul_core(const Rep& a, u32 b, const Rep& a2, u32 b2, const Lane& ln) {
  u64 acc0 = col_ab(a, b, 0);
  u64 acc1 = mad64(a.l[8], row_ror<8>(b), 0);              // lane 0: column 16 (lane 1: a_8 b_9 = 0; lanes >= 2 unused)
  if constexpr (TWO) {
    acc0 = col_ab(a2, b2, acc0);
    acc1 = mad64(a2.l[8], row_ror<8>(b2), acc1);
  }
  const u32 low = recut(acc0) & ln.lt9;                    // == product mod R, limbs < 2^30 + 64
  const u32 m = recut(col_const<RP::NPRIME>(low, 0)) & ln.lt9;      // == -product / p mod R, m < 2.01 R
  acc0 = col_const<P::P>(m, acc0);
  acc1 = mad64(P::P[8], row_ror<8>(m), acc1);
  // product + m p is divisible by R: the result sits in columns 9..17, plus the carry e of the low half, e = (n_8 + 4) >> 29
  const u32 lo0 = (u32)acc0 & MASK29, mid0 = (u32)(acc0 >> W29) & MASK29, hi0 = (u32)(acc0 >> (2 * W29));
  const u32 lo1 = (u32)acc1 & MASK29, mid1 = (u32)(acc1 >> W29) & MASK29;
  const u32 n = lo0 + row_shr<1>(mid0) + row_shr<2>(hi0);
  const u32 e = row_shl<8>((n + 4u) >> W29);               // lane 0 <- lane 8
  u32 r = row_shl<9>(lo0) + row_shl<8>(mid0) + row_shl<7>(hi0) + row_shr<7>(lo1) + row_shr<8>(mid1);
  r += (ln.j == 0) ? e : 0u;
  return r & ln.lt9;
}
__device__ __forceinline__ u32 mul(u32 a_d, u32 b_d, const Lane& ln) {
  const Rep a = rep(a_d);
  return mul_core<false>(a, b_d, a, b_d, ln);
}
__device__ __forceinline__ u32 mul2(u32 a_d, u32 b_d, u32 a2_d, u32 b2_d, const Lane& ln) {
  return mul_core<true>(rep(a_d), b_d, rep(a2_d), b2_d, ln);
}

// lazy normalisation: any limbs < 2^32 -> limbs < 2^29 + 8 (top limb free), same value, one parallel carry step
__device__ __forceinline__ u32 norm(u32 x, const Lane& ln) {
  const u32 lo = (ln.j < 8) ? (x & MASK29) : x;
  const u32 c = (x >> W29) & ln.lt8;
  return lo + row_shr<1>(c);
}
// a - b + K p, lazily normalised.  b lazy (limbs < 2^30 + 64), b < (K - 0.5) p.
template <int K> __device__ __forceinline__ u32 sub(u32 a, u32 b, const Lane& ln) { return norm(a + (kps<K>(ln) - b), ln); }
template <int K> __device__ __forceinline__ u32 neg(u32 b, const Lane& ln) { return norm(kps<K>(ln) - b, ln); }

// exact normalisation: limbs < 2^32 in -> limbs < 2^29 (top limb free).  One parallel carry step, then the remaining 0/1
// carries of all four rows at once from the generate / propagate masks (carry into lane j = bit j of (A + B) ^ A ^ B with
// A = generate | propagate, B = generate; the zero limbs 9..15 of every row stop a carry at the row's end).
__device__ __forceinline__ u32 exact(u32 x, const Lane& ln) {
  const u32 y = norm(x, ln);                                           // limbs j < 8 now <= 2^29 + 6
  const bool g = (ln.j < 8) && y > MASK29, pr = (ln.j < 8) && y == MASK29;
  const u64 G = __builtin_amdgcn_ballot_w64(g), Pm = __builtin_amdgcn_ballot_w64(pr);
  const u64 A = G | Pm;
  const u64 cin = ((A + G) ^ A ^ G);
  const u32 c = (u32)((cin >> (threadIdx.x & 63)) & 1ull);
  const u32 z = y + c;
  return (ln.j < 8) ? (z & MASK29) : z;
}

// x == 0 (mod p) is only possible if the low limb says x = k p for a small k: x_0 p_0^-1 = k (mod 2^29).  x: the RAW limb-wise
// sum (lane 0 never receives a carry, so its low 29 bits are exact), value < (KMAX + 1) p.  Wave-uniform answer (all rows hold
// the same element).
template <int KMAX> __device__ __forceinline__ bool maybe_zero_mod_p(u32 x_raw, const Lane& ln) {
  const u32 k = ((x_raw & MASK29) * RP::PINV29) & MASK29;
  return __builtin_amdgcn_ballot_w64(ln.j == 0 && k <= (u32)KMAX) != 0;
}

// ---- points -------------------------------------------------------------------------------------------------------------
struct Pt { u32 X, Y, ZZ, ZZZ; };            // distributed coordinates, the same copy in every row of the wave
__device__ __forceinline__ bool is_inf(const Pt& p) { return __builtin_amdgcn_ballot_w64(p.ZZ != 0) == 0; }
__device__ __forceinline__ Pt pt_inf() { return Pt{0, 0, 0, 0}; }

// packed record (32 words: X | Y | ZZ | ZZZ, 8 words each) -> distributed limbs.  rec may point to global or LDS memory.
__device__ __forceinline__ u32 load_coord(const u32* __restrict__ w, const Lane& ln) {
  const u32 lo = w[ln.wk], hi = w[(ln.wk + 1) & 7] & ln.hi_ok;
  const u32 v = __builtin_amdgcn_alignbit(hi, lo, (u32)ln.ws);
  return ((ln.j < 8) ? (v & MASK29) : v) & ln.lt9;
}
__device__ __forceinline__ Pt load(const u32* __restrict__ rec, const Lane& ln) {
  Pt p;
  p.X = load_coord(rec, ln); p.Y = load_coord(rec + 8, ln); p.ZZ = load_coord(rec + 16, ln); p.ZZZ = load_coord(rec + 24, ln);
  return p;
}
// row r packs coordinate r: exact limbs, word w = (l_w >> 3 w) | (l_(w+1) << (29 - 3 w)); 32 lanes store one dword each
__device__ __forceinline__ void store(u32* __restrict__ rec, const Pt& p, const Lane& ln) {
  const u32 l = exact(sel4(ln, p.X, p.Y, p.ZZ, p.ZZZ), ln);
  const u32 nxt = row_shl<1>(l);
  const u32 word = (l >> (3 * ln.j)) | (nxt << ((29 - 3 * ln.j) & 31));
  if (ln.j < 8) rec[8 * ln.row + ln.j] = is_inf(p) ? 0u : word;
}

// the rare exceptional pair: hand both points to the plain formulas (same instruction stream in every lane, scalar operands)
__device__ __forceinline__ Fq to_scalar(u32 d) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (u32)__builtin_amdgcn_readlane((int)d, i);
  return fe_carry<P>(r);
}
__device__ __forceinline__ u32 from_scalar(const Fq& v, const Lane& ln) {
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) r = (ln.j == i) ? v.l[i] : r;
  return r;
}
__device__ __noinline__ Pt add_slow(const Pt& a, const Pt& b, const Lane& ln) {
  Xyzz x, y;
  x.X = to_scalar(a.X); x.Y = to_scalar(a.Y); x.ZZ = to_scalar(a.ZZ); x.ZZZ = to_scalar(a.ZZZ);
  y.X = to_scalar(b.X); y.Y = to_scalar(b.Y); y.ZZ = to_scalar(b.ZZ); y.ZZZ = to_scalar(b.ZZZ);
  const Xyzz s = xyzz_add(x, y);
  if (xyzz_is_inf(s)) return pt_inf();
  return Pt{from_scalar(s.X, ln), from_scalar(s.Y, ln), from_scalar(s.ZZ, ln), from_scalar(s.ZZZ, ln)};
}

// a + b (add-2008-s), exception-complete (curve.rs:104-115).  Inputs < 2.5; outputs X3 < 2.07, Y3 < 2.41, ZZ3, ZZZ3 < 2.05.
__device__ __forceinline__ Pt add(const Pt& a, const Pt& b, const Lane& ln) {
  if (is_inf(a)) return b;
  if (is_inf(b)) return a;
  // level 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1          (each < 2.05)
  u32 U1, U2, S1, S2;
  gather4(mul(sel4(ln, a.X, b.X, a.Y, b.Y), sel4(ln, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ), ln), U1, U2, S1, S2);
  const u32 p_raw = U2 + (kps<3>(ln) - U1), r_raw = S2 + (kps<3>(ln) - S1);
  if (maybe_zero_mod_p<5>(p_raw, ln)) return add_slow(a, b, ln);            // P == +-Q (or a one-in-2^26 false alarm)
  const u32 Pd = norm(p_raw, ln), Rd = norm(r_raw, ln);                      // < 5.05
  // level 2: PP = P^2 | RR = R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2                        (PP, RR < 2.17)
  u32 PP, RR, ZZp, ZZZp;
  gather4(mul(sel4(ln, Pd, Rd, a.ZZ, a.ZZZ), sel4(ln, Pd, Rd, b.ZZ, b.ZZZ), ln), PP, RR, ZZp, ZZZp);
  // level 3: PPP = P PP | Q = U1 PP | ZZ3 = (ZZ1 ZZ2) PP | ZZZ1 ZZZ2 again (unused)      (PPP < 2.08, Q < 2.04)
  u32 PPP, Q, ZZ3, unused;
  gather4(mul(sel4(ln, Pd, U1, ZZp, ZZp), PP, ln), PPP, Q, ZZ3, unused);
  const u32 X3raw = sub<7>(RR, norm(PPP + 2u * Q, ln), ln);                  // RR - PPP - 2 Q + 7 p < 9.17
  const u32 Vd = sub<10>(Q, X3raw, ln);                                      // Q - X3 + 10 p < 12.04
  const u32 nS1 = neg<3>(S1, ln);                                            // 3 p - S1 < 3
  // level 4: Y3 = R Vd + (3 p - S1) PPP (one reduction) | X3 = X3raw * (R mod p) | ZZZ3 = (ZZZ1 ZZZ2) PPP
  u32 Y3, X3, ZZZ3;
  gather4(mul2(sel4(ln, Rd, ln.one, ZZZp, ZZZp), sel4(ln, Vd, X3raw, PPP, PPP), nS1, (ln.row == 0) ? PPP : 0u, ln), Y3, X3, ZZZ3, unused);
  return Pt{X3, Y3, ZZ3, ZZZ3};
}

// 2 p (dbl-2008-s-1, a = 0).  Input < 2.5; outputs X3 < 2.06, Y3 < 2.41, ZZ3, ZZZ3 < 2.05.  No exceptional case: the group
// has odd order, so Y != 0 for every finite point.
__device__ __forceinline__ Pt dbl(const Pt& p, const Lane& ln) {
  if (is_inf(p)) return p;
  const u32 U = norm(p.Y << 1, ln);                                          // < 5
  // level 1: V = U^2 | X^2                                                    (V < 2.16, X2 < 2.05)
  u32 V, X2, u0, u1;
  gather4(mul(sel4(ln, U, p.X, U, p.X), sel4(ln, U, p.X, U, p.X), ln), V, X2, u0, u1);
  const u32 M = norm(X2 * 3u, ln);                                           // < 6.15
  // level 2: W = U V | S = X V | MM = M^2 | ZZ3 = V ZZ                        (W < 2.07, S < 2.04, MM < 2.23)
  u32 Wd, S, MM, ZZ3;
  gather4(mul(sel4(ln, U, p.X, M, V), sel4(ln, V, V, M, p.ZZ), ln), Wd, S, MM, ZZ3);
  const u32 X3raw = sub<5>(MM, norm(S << 1, ln), ln);                        // MM - 2 S + 5 p < 7.23
  const u32 Vd = sub<8>(S, X3raw, ln);                                       // S - X3 + 8 p < 10.04
  const u32 nY = neg<3>(p.Y, ln);                                            // 3 p - Y1 < 3
  // level 3: Y3 = M Vd + (3 p - Y1) W | X3 = X3raw * (R mod p) | ZZZ3 = W ZZZ
  u32 Y3, X3, ZZZ3;
  gather4(mul2(sel4(ln, M, ln.one, Wd, Wd), sel4(ln, Vd, X3raw, p.ZZZ, p.ZZZ), nY, (ln.row == 0) ? Wd : 0u, ln), Y3, X3, ZZZ3, u0);
  return Pt{X3, Y3, ZZ3, ZZZ3};
}

}  // namespace rowop
}  // namespace mzk
