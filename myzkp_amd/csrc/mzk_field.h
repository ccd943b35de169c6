// mzk_field.h -- prime-field arithmetic for the MyZKP prover hot path on gfx950.
//
// Replaces, on the device, the BigInt-backed FiniteFieldElement<M> of the reference
// (myzkp/src/modules/algebra/field.rs:87-133 type, :157-207 Ring impl, :209-279 Field impl) for the
// three moduli on the path: Fr = ModEIP197 (field.rs:428-431), Fq = BN128Modulus
// (curve/bn128.rs:19-22), M128 (zkstark/fri.rs:408).
//
// Representation: unsaturated 29-bit limbs (L = 9 for the 254-bit fields, 5 for M128), Montgomery
// radix R = 2^(29 L).  Why 29 bits: on gfx950 v_mad_u64_u32 (32x32+64 -> 64) issues at half rate and
// takes its 64-bit addend for free, so a product-scanning column sum  col += a_i * b_j  costs exactly
// one instruction and no carry handling as long as a column never exceeds 64 bits; 18 terms of
// 29x29 bits do not.  Measured on MI355X (profiles/r01_ubench_instr_rates.txt, tools/microbench/mm29.hip):
// 154 G Montgomery mul/s in this form vs 93 G/s for saturated 8x32 CIOS (carry chains: 300 v_mov +
// 138 v_lshl_add_u64 per product).
//
// The header is plain C++ so that the same code compiles with g++ for the bounds-checked host unit
// tests (tests/hostcheck); the shipped library only instantiates it inside HIP kernels.
//
// Invariants (checked in host builds with -DMZK_CHECK_BOUNDS):
//   normalised  : l[i] < 2^29 for i < L-1 (top limb free).
//   fe_mul(a,b) : needs max_limb(a) * max_limb(b) * L + L * 2^58 < 2^64  (e.g. both < 2^30, or
//                 2^30.6 x 2^29); returns a normalised value  < a*b/R + p.
//   Values are only ever compared after fe_reduce() (canonical representative in [0,p)), which is
//   the reference's own parity definition (sanitize(), field.rs:260-270; PartialEq field.rs:290-294).
#pragma once
#include <stdint.h>
#include "mzk_constants.h"

#if defined(__HIPCC__)
#define MZK_HD __host__ __device__ __forceinline__
#else
#define MZK_HD inline
#endif

#if defined(MZK_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#define MZK_ASSERT(x) assert(x)
#else
#define MZK_ASSERT(x) ((void)0)
#endif

namespace mzk {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

// col + a * b: one v_mad_u64_u32.  hipcc starts every column of a product from zero and adds the previous column's carry
// with an extra v_lshl_add_u64 (instruction-level parallelism for a lone wave); the throughput kernels use the
// hand-scheduled single-chain forms of mzk_field_asm.h instead (FeAsm: 214 instead of ~255 instructions per product).
MZK_HD uint64_t mzk_mad(uint32_t a, uint32_t b, uint64_t c) { return c + (uint64_t)a * b; }

constexpr int W29 = 29;
constexpr u32 MASK29 = (1u << 29) - 1u;

template <class P> struct Fe { u32 l[P::L]; };

template <class P> MZK_HD Fe<P> fe_zero() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = 0;
  return r;
}
// Montgomery form of 1.
template <class P> MZK_HD Fe<P> fe_one() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = P::ONE[i];
  return r;
}
template <class P> MZK_HD Fe<P> fe_r2() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = P::R2[i];
  return r;
}

// ---------------------------------------------------------------------------------------------
// Sparse modulus p = PT * 2^(29 (L-1)) + 1 (M128 = 1 + 407 * 2^119 = 3256 * 2^116 + 1, fri.rs:408): the Montgomery reduction
// needs no multiplication by -p^-1 and one multiply-add per limb.  With beta = 2^29 and T = a b = T_lo + beta^(L-1) T_hi
// (T_lo = the L-1 carried low limbs t_k), adding (beta^(L-1) - T_lo) p clears the low limbs:
//     T beta^-(L-1)  ==  U = T_hi + PT * sum_k (MASK - t_k) beta^k + (PT + 1)           (mod p),
// and one more limb the same way, U = u0 + beta U_hi, m = (1 + C) beta - u0:
//     T beta^-L      ==  V = U_hi + (1 + C) + PT * m * beta^(L-2)  =  (T + (..) p) / R   [+ C p].
// 25 + 5 multiply-adds for L = 5 and no v_mul_lo, against 25 + 10 + 5; the complements MASK - t_k are one v_bfi each.
// Result: limbs 0 .. L-2 in [0, 2^29), top limb whatever is left; value in [T / R + C p, T / R + (1 + C) p + p / 2^29 + 3).
// SIGNED: the limbs of `a` are two's-complement i32 (|a_i| < 2^31; the lazily accumulated butterflies of the NTT), columns are
// signed 64-bit sums (v_mad_i64_i32, arithmetic shifts: floor semantics throughout), the top limb of the result is signed;
// `b` always has non-negative limbs below 2^29.  A non-negative `a` gives a non-negative result.
// ---------------------------------------------------------------------------------------------
template <class P> struct SparseMod {
  static constexpr bool value = (P::L == 5) && P::P[0] == 1 && P::P[1] == 0 && P::P[2] == 0 && P::P[3] == 0;
};
template <bool S> struct AccOf { typedef u64 acc; typedef u32 limb; };
template <> struct AccOf<true> { typedef int64_t acc; typedef i32 limb; };
template <class P, bool SIGNED, int C> MZK_HD Fe<P> fe_mul_sparse(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L;
  static_assert(SparseMod<P>::value, "p = PT 2^(29 (L-1)) + 1");
  typedef typename AccOf<SIGNED>::acc A;
  typedef typename AccOf<SIGNED>::limb LL;
  constexpr u32 PT = P::P[L - 1];
  u32 n[L - 1];
  Fe<P> r;
  A acc = 0;
#if defined(MZK_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
  {
    __int128 ma = 0;
    for (int i = 0; i < L; i++) {
      const __int128 v = SIGNED ? (__int128)(i32)a.l[i] : (__int128)a.l[i];
      if ((v < 0 ? -v : v) > ma) ma = v < 0 ? -v : v;
      assert(b.l[i] <= MASK29 || i == L - 1);
    }
    const __int128 worst = ma * MASK29 * L + (__int128)PT * ((__int128)2 << 30) + ((__int128)1 << 36);
    assert(worst < ((__int128)1 << (SIGNED ? 63 : 64)));
  }
#endif
#pragma unroll
  for (int k = 0; k < L - 1; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (A)(LL)a.l[i] * (A)(LL)b.l[k - i];
    n[k] = ~(u32)acc & MASK29;
    acc >>= W29;
  }
#pragma unroll
  for (int k = L - 1; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) acc += (A)(LL)a.l[i] * (A)(LL)b.l[k - i];
    if (k < 2 * L - 2) acc += (A)((u64)PT * n[k - L + 1]);
    if (k == L - 1) {        // U's lowest limb: the constants of both steps enter here, then the second step's multiplier
      acc += (A)(u64)(PT + 1 + ((u32)(1 + C) << W29));
      n[0] = ((u32)(1 + C) << W29) - ((u32)acc & MASK29);
    } else {
      if (k == 2 * L - 2) acc += (A)((u64)PT * n[0]);
      r.l[k - L] = (u32)acc & MASK29;
    }
    acc >>= W29;
  }
  MZK_ASSERT(SIGNED ? ((int64_t)acc >= -((int64_t)1 << 31) && (int64_t)acc < ((int64_t)1 << 31)) : ((u64)acc < ((u64)1 << 32)));
  r.l[L - 1] = (u32)acc;
  return r;
}

// ---- signed lazy limbs (the NTT butterflies of a sparse-modulus field, mzk_ntt.hip) -------------------------------------------
// A value is sum l_i beta^i with l_i read as i32: sums and differences are ONE instruction per limb (no K p, no carry);
// fe_scarry brings limbs 0 .. L-2 back into [0, 2^29) (the top limb stays signed), the signed product does the same as a side effect.
template <class P> MZK_HD Fe<P> fe_sadd(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    MZK_ASSERT((int64_t)(i32)a.l[i] + (i32)b.l[i] < ((int64_t)1 << 31) && (int64_t)(i32)a.l[i] + (i32)b.l[i] >= -((int64_t)1 << 31));
    r.l[i] = a.l[i] + b.l[i];
  }
  return r;
}
template <class P> MZK_HD Fe<P> fe_ssub(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    MZK_ASSERT((int64_t)(i32)a.l[i] - (i32)b.l[i] < ((int64_t)1 << 31) && (int64_t)(i32)a.l[i] - (i32)b.l[i] >= -((int64_t)1 << 31));
    r.l[i] = a.l[i] - b.l[i];
  }
  return r;
}
template <class P> MZK_HD Fe<P> fe_scarry(const Fe<P>& a) {
  Fe<P> r;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < P::L - 1; i++) {
    MZK_ASSERT((int64_t)(i32)a.l[i] + c < ((int64_t)1 << 31) && (int64_t)(i32)a.l[i] + c >= -((int64_t)1 << 31));
    const i32 v = (i32)a.l[i] + c;
    r.l[i] = (u32)v & MASK29;
    c = v >> W29;        // arithmetic
  }
  MZK_ASSERT((int64_t)(i32)a.l[P::L - 1] + c < ((int64_t)1 << 31) && (int64_t)(i32)a.l[P::L - 1] + c >= -((int64_t)1 << 31));
  r.l[P::L - 1] = a.l[P::L - 1] + (u32)c;
  return r;
}
// Signed lazy -> the same residue as a NON-NEGATIVE lazy value: + SBIAS p limb-wise (three additions).  |value| < SBIAS p.
template <class P> struct SLazy {
  static constexpr u32 BIAS = 1u << 12;       // covers 2^10 levels of doubling on 128-bit inputs: |x| < 2^138 < 2^12 p
};
template <class P> MZK_HD Fe<P> fe_sbias(const Fe<P>& a) {
  static_assert(SparseMod<P>::value, "p = PT 2^(29 (L-1)) + 1");
  Fe<P> x = a;
  // limb 0's share of the bias goes in as BIAS - 2^29 with the 2^29 carried into limb 1 by hand: limbs of 3 (2^29 - 1) stay inside
  // i32 through the carry pass that follows in fe_sreduce
  MZK_ASSERT((int64_t)(i32)a.l[0] + SLazy<P>::BIAS - ((int64_t)1 << W29) >= -((int64_t)1 << 31) && (int64_t)(i32)a.l[1] + 1 < ((int64_t)1 << 31));
  x.l[0] += SLazy<P>::BIAS - (1u << W29);
  x.l[1] += 1u;
  x.l[P::L - 1] += P::P[P::L - 1] * SLazy<P>::BIAS;
  return x;
}
// Canonical representative in [0, p) of a signed lazy value, |value| < SBIAS p, limbs within +-3 (2^29 - 1) (what the NTT's stage pairs leave).  After the bias and one carry
// pass the top limb holds floor(x / 2^116) < 2^26; with x = x_lo + 2^116 (PT q + r) and 2^116 PT == -1 the residue is
// x_lo + 2^116 r - q: canonical as it stands whenever limb 0 >= q (q < 2^14: all but ~2^-15 of the inputs); the rest borrows.
template <class P> MZK_HD Fe<P> fe_sreduce(const Fe<P>& a) {
  static_assert(SparseMod<P>::value, "p = PT 2^(29 (L-1)) + 1");
  constexpr int L = P::L;
  constexpr u32 PT = P::P[L - 1];
  constexpr u64 QM = (((u64)1 << 43) + PT - 1) / PT;      // floor(t / PT) = (t QM) >> 43 exactly for t < 2^26 (t (QM PT - 2^43) < 2^43)
  static_assert(QM < ((u64)1 << 32) && PT < (1u << 12), "quotient constant");
  Fe<P> x = fe_scarry<P>(fe_sbias<P>(a));
  MZK_ASSERT((i32)x.l[L - 1] >= 0 && x.l[L - 1] < (1u << 26));
  const u32 q = (u32)(((u64)x.l[L - 1] * QM) >> 43);
  const u32 r = x.l[L - 1] - q * PT;
  MZK_ASSERT(r < PT);
  if (x.l[0] >= q) {
    x.l[0] -= q;
    x.l[L - 1] = r;
    return x;
  }
  x.l[0] -= q;
  x.l[L - 1] = r;
  x = fe_scarry<P>(x);
  if ((i32)x.l[L - 1] < 0) {
    x.l[0] += 1;
    x.l[L - 1] += PT;
    x = fe_scarry<P>(x);
  }
  return x;
}

// ---------------------------------------------------------------------------------------------
// Montgomery product  a*b/R mod p  (finely integrated product scanning, one 64-bit column
// accumulator).  Reference semantics: Ring::mul_ref, field.rs:176-179 (value * value % modulus).
// ---------------------------------------------------------------------------------------------
template <class P> MZK_HD Fe<P> fe_mul_dense(const Fe<P>& a, const Fe<P>& b);
template <class P> MZK_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
  if constexpr (SparseMod<P>::value) return fe_mul_sparse<P, false, 0>(a, b);
  else return fe_mul_dense<P>(a, b);
}
template <class P> MZK_HD Fe<P> fe_mul_dense(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L;
  u32 m[L];
  Fe<P> r;
  u64 col = 0;
#if defined(MZK_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
  {  // the widest column must fit 64 bits
    unsigned __int128 worst = 0;
    u64 ma = 0, mb = 0;
    for (int i = 0; i < L; i++) { if (a.l[i] > ma) ma = a.l[i]; if (b.l[i] > mb) mb = b.l[i]; }
    worst = (unsigned __int128)ma * mb * L + (unsigned __int128)L * ((u64)MASK29 * MASK29) + ((u64)1 << 36);
    assert(worst < ((unsigned __int128)1 << 64));
  }
#endif
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = 0; i < k; i++) if (P::P[k - i] != 0) col = mzk_mad(m[i], P::P[k - i], col);
    m[k] = ((u32)col * P::N0) & MASK29;
    if (P::P[0] != 0) col = mzk_mad(m[k], P::P[0], col);
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) if (P::P[k - i] != 0) col = mzk_mad(m[i], P::P[k - i], col);
    r.l[k - L] = (u32)col & MASK29;
    col >>= W29;
  }
  MZK_ASSERT(col < ((u64)1 << 32));
  r.l[L - 1] = (u32)col;
  return r;
}

// Precomputed-quotient (Shoup) product for a CONSTANT factor: x * w mod p with plain w (limbs w[L]) and
// wq = floor(w 2^(29 L) / p) (limbs wq[L]) -- no Montgomery form anywhere.  q = columns L .. 2L-1 of x * wq (columns L-2, L-1
// computed as guards, the lower ones dropped: |q - floor(x w / p)| <= 2), r = low L columns of x * w + q (2^(29 L) - p).
// r == x w (mod p), r < 4 p (measured: < 1.3 p), limbs normalised.  x may be lazy: limbs up to 3 * 2^30, value < 2^(29 L).
// 143 multiply-adds against the Montgomery product's 162 for L = 9 (tools/shoup_model.py is the integer model; the NTT uses
// it where a whole wave shares a twiddle and w, wq sit in scalar registers).
template <class P> MZK_HD Fe<P> fe_shoup_mul(const Fe<P>& x, const u32* w, const u32* wq) {
  constexpr int L = P::L;
  u32 q[L];
  Fe<P> r;
  u64 col = 0;
#if defined(MZK_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
  {
    u64 mx = 0;
    for (int i = 0; i < L; i++) { if (x.l[i] > mx) mx = x.l[i]; assert(w[i] <= MASK29 && wq[i] <= MASK29); }
    const unsigned __int128 worst = (unsigned __int128)mx * MASK29 * L + (unsigned __int128)L * ((u64)MASK29 * MASK29) + ((u64)1 << 36);
    assert(worst < ((unsigned __int128)1 << 64));
  }
#endif
#pragma unroll
  for (int k = L - 2; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = (k - L + 1 > 0 ? k - L + 1 : 0); i <= (k < L - 1 ? k : L - 1); i++) col = mzk_mad(x.l[i], wq[k - i], col);
    if (k >= L) q[k - L] = (u32)col & MASK29;
    col >>= W29;
  }
  MZK_ASSERT(col < ((u64)1 << 32));
  q[L - 1] = (u32)col;
  col = 0;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) col = mzk_mad(x.l[i], w[k - i], col);
#pragma unroll
    for (int i = 0; i <= k; i++) col = mzk_mad(q[i], P::RMP[k - i], col);
    r.l[k] = (u32)col & MASK29;
    col >>= W29;
  }
  return r;
}

// a*b + c*d with ONE Montgomery reduction (the two double-width products share the 64-bit columns):
// (a b + c d)/R mod p, normalised, < (a b + c d)/R + p.  Saves a whole reduction (81 MADs + 9 v_mul_lo)
// wherever a formula adds or subtracts two products.  Needs
//   L (max_limb(a) max_limb(b) + max_limb(c) max_limb(d)) + L 2^58 < 2^64.
template <class P> MZK_HD Fe<P> fe_mul_add2(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) {
  constexpr int L = P::L;
  u32 m[L];
  Fe<P> r;
  u64 col = 0;
#if defined(MZK_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
  {
    u64 ma = 0, mb = 0, mc = 0, md = 0;
    for (int i = 0; i < L; i++) {
      if (a.l[i] > ma) ma = a.l[i]; if (b.l[i] > mb) mb = b.l[i];
      if (c.l[i] > mc) mc = c.l[i]; if (d.l[i] > md) md = d.l[i];
    }
    unsigned __int128 worst = ((unsigned __int128)ma * mb + (unsigned __int128)mc * md) * L +
                              (unsigned __int128)L * ((u64)MASK29 * MASK29) + ((u64)1 << 36);
    assert(worst < ((unsigned __int128)1 << 64));
  }
#endif
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = 0; i <= k; i++) col = mzk_mad(c.l[i], d.l[k - i], col);
#pragma unroll
    for (int i = 0; i < k; i++) if (P::P[k - i] != 0) col = mzk_mad(m[i], P::P[k - i], col);
    m[k] = ((u32)col * P::N0) & MASK29;
    if (P::P[0] != 0) col = mzk_mad(m[k], P::P[0], col);
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col = mzk_mad(a.l[i], b.l[k - i], col);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) col = mzk_mad(c.l[i], d.l[k - i], col);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) if (P::P[k - i] != 0) col = mzk_mad(m[i], P::P[k - i], col);
    r.l[k - L] = (u32)col & MASK29;
    col >>= W29;
  }
  MZK_ASSERT(col < ((u64)1 << 32));
  r.l[L - 1] = (u32)col;
  return r;
}
// K p - b limb-wise (borrow-friendly limbs, no carries): a non-negative representative of -b with limbs
// < 2^30.6; b normalised with value < (K/2) p.
template <class P, int K> MZK_HD Fe<P> fe_neg_lazy(const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    const u32 c = (K == 2) ? P::KP2[i] : (K == 4) ? P::KP4[i] : (K == 8) ? P::KP8[i] : P::KP16[i];
    MZK_ASSERT(c >= b.l[i]);
    r.l[i] = c - b.l[i];
  }
  return r;
}

// Montgomery square: cross products once, doubled (45 instead of 81 product terms for L = 9).
template <class P> MZK_HD Fe<P> fe_sqr(const Fe<P>& a) {
  constexpr int L = P::L;
  u32 m[L], a2[L];
  Fe<P> r;
  u64 col = 0;
#pragma unroll
  for (int i = 0; i < L; i++) a2[i] = a.l[i] << 1;
#if defined(MZK_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
  {
    u64 ma = 0;
    for (int i = 0; i < L; i++) { assert(a.l[i] < (1u << 31)); if (a.l[i] > ma) ma = a.l[i]; }
    unsigned __int128 worst = (unsigned __int128)ma * ma * L + (unsigned __int128)L * ((u64)MASK29 * MASK29) + ((u64)1 << 36);
    assert(worst < ((unsigned __int128)1 << 64));
  }
#endif
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) col = mzk_mad(a2[i], a.l[k - i], col);
    if ((k & 1) == 0) col = mzk_mad(a.l[k / 2], a.l[k / 2], col);
#pragma unroll
    for (int i = 0; i < k; i++) if (P::P[k - i] != 0) col = mzk_mad(m[i], P::P[k - i], col);
    m[k] = ((u32)col * P::N0) & MASK29;
    if (P::P[0] != 0) col = mzk_mad(m[k], P::P[0], col);
    col >>= W29;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; 2 * i < k; i++) col = mzk_mad(a2[i], a.l[k - i], col);
    if ((k & 1) == 0) col = mzk_mad(a.l[k / 2], a.l[k / 2], col);
#pragma unroll
    for (int i = k - L + 1; i < L; i++) if (P::P[k - i] != 0) col = mzk_mad(m[i], P::P[k - i], col);
    r.l[k - L] = (u32)col & MASK29;
    col >>= W29;
  }
  MZK_ASSERT(col < ((u64)1 << 32));
  r.l[L - 1] = (u32)col;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Lazy additive ops: limb-wise, no carry propagation, no modular reduction.
// ---------------------------------------------------------------------------------------------
template <class P> MZK_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    MZK_ASSERT((u64)a.l[i] + b.l[i] < ((u64)1 << 32));
    r.l[i] = a.l[i] + b.l[i];
  }
  return r;
}
// a - b + K p  (K in {2,4,8,16}); b must be normalised with value < (K/2) p.
template <class P, int K> MZK_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    const u32 c = (K == 2) ? P::KP2[i] : (K == 4) ? P::KP4[i] : (K == 8) ? P::KP8[i] : P::KP16[i];
    MZK_ASSERT(c >= b.l[i]);
    MZK_ASSERT((u64)a.l[i] + (c - b.l[i]) < ((u64)1 << 32));
    r.l[i] = a.l[i] + (c - b.l[i]);
  }
  return r;
}
template <class P> MZK_HD Fe<P> fe_dbl(const Fe<P>& a) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    MZK_ASSERT(a.l[i] < (1u << 31));
    r.l[i] = a.l[i] << 1;
  }
  return r;
}
// Carry propagation: any limbs -> normalised limbs, same value.
template <class P> MZK_HD Fe<P> fe_carry(const Fe<P>& a) {
  Fe<P> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::L - 1; i++) {
    MZK_ASSERT((u64)a.l[i] + c < ((u64)1 << 32));
    u32 v = a.l[i] + c;
    r.l[i] = v & MASK29;
    c = v >> W29;
  }
  MZK_ASSERT((u64)a.l[P::L - 1] + c < ((u64)1 << 32));
  r.l[P::L - 1] = a.l[P::L - 1] + c;
  return r;
}

// Fused (a + b) and (a - b + K p) with carry propagation: normalised result, one pass over the limbs.
template <class P> MZK_HD Fe<P> fe_add_carry(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::L - 1; i++) {
    MZK_ASSERT((u64)a.l[i] + b.l[i] + c < ((u64)1 << 32));
    const u32 v = a.l[i] + b.l[i] + c;
    r.l[i] = v & MASK29;
    c = v >> W29;
  }
  MZK_ASSERT((u64)a.l[P::L - 1] + b.l[P::L - 1] + c < ((u64)1 << 32));
  r.l[P::L - 1] = a.l[P::L - 1] + b.l[P::L - 1] + c;
  return r;
}
template <class P, int K> MZK_HD Fe<P> fe_sub_carry(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    const u32 kp = (K == 2) ? P::KP2[i] : (K == 4) ? P::KP4[i] : (K == 8) ? P::KP8[i] : P::KP16[i];
    MZK_ASSERT(kp >= b.l[i]);
    MZK_ASSERT((u64)a.l[i] + (kp - b.l[i]) + c < ((u64)1 << 32));
    const u32 v = a.l[i] + (kp - b.l[i]) + c;
    if (i < P::L - 1) { r.l[i] = v & MASK29; c = v >> W29; }
    else r.l[i] = v;
  }
  return r;
}

// x - p if that is >= 0, else x.  x must be normalised.
template <class P> MZK_HD Fe<P> fe_cond_sub_p(const Fe<P>& x) {
  constexpr int L = P::L;
  Fe<P> d;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < L - 1; i++) {
    i32 v = (i32)x.l[i] - (i32)P::P[i] + c;
    d.l[i] = (u32)v & MASK29;
    c = v >> W29;  // arithmetic: 0 or -1
  }
  i32 top = (i32)x.l[L - 1] - (i32)P::P[L - 1] + c;
  d.l[L - 1] = (u32)top;
  const bool neg = top < 0;
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.l[i] = neg ? x.l[i] : d.l[i];
  return r;
}

// Canonical representative in [0, p).  Accepts any limbs whose carried top limb stays below
// P::TOPMAX (every value the kernels produce is far below that).  Mirrors Field::sanitize
// (field.rs:260-270).  Quotient estimate q = floor(top * QM / 2^QS) <= floor(top / (Ptop+1))
// <= floor(x / p), and q >= floor(x / p) - 2, so two conditional subtractions finish.
template <class P> MZK_HD Fe<P> fe_weak_reduce(const Fe<P>& a);
template <class P> MZK_HD Fe<P> fe_reduce(const Fe<P>& a) {
  Fe<P> x = fe_weak_reduce<P>(a);
  x = fe_cond_sub_p<P>(x);
  x = fe_cond_sub_p<P>(x);
  return x;
}

// Weak reduction: normalised limbs, same residue, value < 2.01 p.  Input: any limbs whose carried
// top limb is below P::TOPMAX.
template <class P> MZK_HD Fe<P> fe_weak_reduce(const Fe<P>& a) {
  constexpr int L = P::L;
  Fe<P> x = fe_carry<P>(a);
  MZK_ASSERT(x.l[L - 1] < P::TOPMAX);
  const u32 q = (u32)(((u64)x.l[L - 1] * P::QM) >> P::QS);
  u64 acc = 0;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < L - 1; i++) {
    acc += (u64)q * P::P[i];
    i32 v = (i32)x.l[i] - (i32)((u32)acc & MASK29) + c;
    x.l[i] = (u32)v & MASK29;
    c = v >> W29;
    acc >>= W29;
  }
  acc += (u64)q * P::P[L - 1];
  {
    int64_t v = (int64_t)x.l[L - 1] - (int64_t)acc + c;
    MZK_ASSERT(v >= 0 && v < ((int64_t)1 << 31));
    x.l[L - 1] = (u32)v;
  }
  return x;
}

template <class P> MZK_HD bool fe_is_zero_canon(const Fe<P>& a) {
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) acc |= a.l[i];
  return acc == 0;
}
template <class P> MZK_HD bool fe_eq_canon(const Fe<P>& a, const Fe<P>& b) {
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) acc |= a.l[i] ^ b.l[i];
  return acc == 0;
}
// value == 0 (mod p) for a lazily reduced value
template <class P> MZK_HD bool fe_is_zero(const Fe<P>& a) { return fe_is_zero_canon<P>(fe_reduce<P>(a)); }

// value == 0 (mod p)?  x normalised with value <= KMAX * p.  Cheap filter on limb 0: x = k p for some k <= KMAX means
// x_0 = k p_0 (mod 2^29), i.e. k = x_0 * p_0^-1 mod 2^29 (p_0^-1 = -N0): one multiply and one compare whatever KMAX is
// (round 2 compared x_0 with every k p_0: 2 KMAX instructions); full reduction only on a hit (a false hit has probability
// KMAX / 2^29).
template <class P, int KMAX> MZK_HD bool fe_is_zero_mod(const Fe<P>& x) {
  const u32 k = (x.l[0] * ((0u - P::N0) & MASK29)) & MASK29;
  if (k > (u32)KMAX) return false;
  return fe_is_zero_canon<P>(fe_reduce<P>(x));
}

// -x mod p for canonical x (canonical result).  Ring Neg, field.rs:296-303.
template <class P> MZK_HD Fe<P> fe_neg_canon(const Fe<P>& x) {
  constexpr int L = P::L;
  Fe<P> d;
  i32 c = 0;
  u32 nz = 0;
#pragma unroll
  for (int i = 0; i < L; i++) nz |= x.l[i];
#pragma unroll
  for (int i = 0; i < L - 1; i++) {
    i32 v = (i32)P::P[i] - (i32)x.l[i] + c;
    d.l[i] = (u32)v & MASK29;
    c = v >> W29;
  }
  d.l[L - 1] = (u32)((i32)P::P[L - 1] - (i32)x.l[L - 1] + c);
#pragma unroll
  for (int i = 0; i < L; i++) d.l[i] = nz ? d.l[i] : 0u;
  return d;
}

// ---------------------------------------------------------------------------------------------
// ABI encoding <-> limbs.  The ABI carries canonical values as NW little-endian 32-bit words
// (= 4 or 2 u64 limbs), the wire format of examples/sumcheck/src/utils.rs:51-72.
// ---------------------------------------------------------------------------------------------
template <class P> MZK_HD Fe<P> fe_unpack(const u32* w) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    const int o = W29 * i, k = o >> 5, s = o & 31;
    u32 lo = (k < P::NW) ? w[k] : 0u;
    u32 hi = (k + 1 < P::NW) ? w[k + 1] : 0u;
    u32 v = (s == 0) ? lo : ((lo >> s) | (hi << (32 - s)));
    r.l[i] = (i < P::L - 1) ? (v & MASK29) : v;
  }
  // the top limb holds bits [29(L-1), 32 NW): already exact
  return r;
}
// requires normalised limbs and value < 2^(32 NW)
template <class P> MZK_HD void fe_pack(const Fe<P>& a, u32* w) {
#pragma unroll
  for (int k = 0; k < P::NW; k++) {
    // word k holds bits [32k, 32k+32)
    const int lo_l = (32 * k) / W29, lo_s = (32 * k) % W29;
    u64 v = (u64)a.l[lo_l] >> lo_s;
    int have = W29 - lo_s;
    if (lo_l + 1 < P::L) v |= (u64)a.l[lo_l + 1] << have;
    have += W29;
    if (have < 32 && lo_l + 2 < P::L) v |= (u64)a.l[lo_l + 2] << have;
    w[k] = (u32)v;
  }
}

template <class P> MZK_HD Fe<P> fe_to_mont(const Fe<P>& plain) { return fe_mul<P>(plain, fe_r2<P>()); }
template <class P> MZK_HD Fe<P> fe_from_mont(const Fe<P>& mont) {
  Fe<P> one = fe_zero<P>();
  one.l[0] = 1;
  return fe_reduce<P>(fe_mul<P>(mont, one));
}

// a^e for a Montgomery-form a; e given as nw little-endian 32-bit words (MSB-first square-multiply).
// Reference: Ring::pow -> utils.rs:108-137 (LSB-first; same value).
template <class P> MZK_HD Fe<P> fe_pow_words(const Fe<P>& a, const u32* e, int nw) {
  Fe<P> r = fe_one<P>();
  bool started = false;
  for (int k = nw - 1; k >= 0; k--) {
    for (int bit = 31; bit >= 0; bit--) {
      if (started) r = fe_sqr<P>(r);
      if ((e[k] >> bit) & 1u) {
        r = started ? fe_mul<P>(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}
template <class P> MZK_HD Fe<P> fe_pow_u64(const Fe<P>& a, u64 e) {
  u32 w[2] = {(u32)e, (u32)(e >> 32)};
  return fe_pow_words<P>(a, w, 2);
}
// Multiplicative inverse by Fermat (a^(p-2)); 0 -> 0.  Branch-free per lane: the form to use when many
// lanes invert different values at once.  The reference uses extended Euclid (field.rs:210-237); the
// value is the same canonical residue.
template <class P> MZK_HD Fe<P> fe_inv(const Fe<P>& a) {
  u32 e[P::NW];
#pragma unroll
  for (int i = 0; i < P::NW; i++) e[i] = P::PM2[i];
  return fe_pow_words<P>(a, e, P::NW);
}

// ---- binary extended GCD on saturated 32-bit words (helpers) ----------------------------------------
template <int NW> MZK_HD bool w_is_one(const u32* a) {
  u32 acc = a[0] ^ 1u;
#pragma unroll
  for (int i = 1; i < NW; i++) acc |= a[i];
  return acc == 0;
}
template <int NW> MZK_HD bool w_ge(const u32* a, const u32* b) {  // a >= b
  bool ge = true;  // equal so far
#pragma unroll
  for (int i = 0; i < NW; i++) {  // least significant first: a later (more significant) difference overrides
    if (a[i] != b[i]) ge = a[i] > b[i];
  }
  return ge;
}
template <int NW> MZK_HD u32 w_sub(u32* r, const u32* a, const u32* b) {  // r = a - b, returns borrow
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const u64 d = (u64)a[i] - b[i] - br;
    r[i] = (u32)d;
    br = (u32)(d >> 63);
  }
  return br;
}
template <int NW> MZK_HD u32 w_add(u32* r, const u32* a, const u32* b) {  // r = a + b, returns carry
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const u64 s = (u64)a[i] + b[i] + c;
    r[i] = (u32)s;
    c = (u32)(s >> 32);
  }
  return c;
}
template <int NW> MZK_HD void w_shr1(u32* a, u32 top) {  // a = (top:a) >> 1
#pragma unroll
  for (int i = 0; i < NW - 1; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
  a[NW - 1] = (a[NW - 1] >> 1) | (top << 31);
}
// x <- x / 2 mod p (p odd): x even -> x >> 1, else (x + p) >> 1
template <class P> MZK_HD void w_half_mod(u32* x) {
  u32 pw[P::NW];
#pragma unroll
  for (int i = 0; i < P::NW; i++) pw[i] = P::PW[i];
  u32 top = 0;
  if (x[0] & 1u) top = w_add<P::NW>(x, x, pw);
  w_shr1<P::NW>(x, top);
}
// x <- x - y mod p, both in [0, p)
template <class P> MZK_HD void w_sub_mod(u32* x, const u32* y) {
  u32 pw[P::NW];
#pragma unroll
  for (int i = 0; i < P::NW; i++) pw[i] = P::PW[i];
  if (w_sub<P::NW>(x, x, y)) w_add<P::NW>(x, x, pw);
}

// Same inverse by binary extended Euclid on the canonical words.  Data-dependent control flow: faster
// than the Fermat ladder on ONE lane (128 vs 172 us measured; the serial tail of every MSM ends in one
// inversion), slower when 64 lanes diverge -- so only the single-lane tail kernels use it.
template <class P> MZK_HD Fe<P> fe_inv_serial(const Fe<P>& a) {
  constexpr int NW = P::NW;
  u32 u[NW], v[NW], x1[NW], x2[NW];
  {
    Fe<P> c = fe_reduce<P>(a);
    if (fe_is_zero_canon<P>(c)) return fe_zero<P>();
    fe_pack<P>(c, u);
  }
#pragma unroll
  for (int i = 0; i < NW; i++) { v[i] = P::PW[i]; x1[i] = 0; x2[i] = 0; }
  x1[0] = 1;
  // invariants: x1 * a == u, x2 * a == v (mod p); gcd(u, v) = 1
  while (!w_is_one<NW>(u) && !w_is_one<NW>(v)) {
    while (!(u[0] & 1u)) { w_shr1<NW>(u, 0); w_half_mod<P>(x1); }
    while (!(v[0] & 1u)) { w_shr1<NW>(v, 0); w_half_mod<P>(x2); }
    if (w_ge<NW>(u, v)) { w_sub<NW>(u, u, v); w_sub_mod<P>(x1, x2); }
    else { w_sub<NW>(v, v, u); w_sub_mod<P>(x2, x1); }
  }
  u32* r = w_is_one<NW>(u) ? x1 : x2;
  // r = (aR)^-1 = a^-1 R^-1 (plain words); Montgomery form of a^-1 is a^-1 R = mont(r, R^3)
  Fe<P> r3;
#pragma unroll
  for (int i = 0; i < P::L; i++) r3.l[i] = P::R3[i];
  return fe_mul<P>(fe_unpack<P>(r), r3);
}

// ---- inverse by safegcd (Bernstein-Yang divsteps), 30 divsteps per batch ---------------------------------
// The single-lane tails (one inversion ends every MSM / fold) are pure latency, and both other inversions are
// long dependent chains (Fermat: ~380 dependent products; binary Euclid: ~760 dependent multi-word steps).
// Divsteps work on the LOW 32 bits of f, g for 30 steps, collect the steps in a 2x2 integer matrix and apply it
// to the full-width f, g and to the cofactors d, e once per batch: ~20 batches for a 254-bit modulus.
// Signed 30-bit limbs and 32-bit matrix entries, so that every product of the batch updates is one
// v_mad_i64_i32 on this 32-bit ALU (round 1 used 62-bit limbs and __int128 products -- about sixteen instructions
// each here: ~2.4 x the instructions for the same inversion).  Variable time (ctz skips zero runs): single lane only.
// Invariants: d x == f, e x == g (mod p) up to the common power of two that the batch update divides out
// with the p^-1 mod 2^30 trick; d, e stay in (-2p, p).
template <int NL> struct Sgn30 { i32 v[NL]; };
struct DivMat { i32 u, v, q, r; };
MZK_HD int sg_ctz32(u32 x) { return __builtin_ctz(x); }
MZK_HD i32 sg_divsteps_30(i32 eta, u32 f0, u32 g0, DivMat* t) {
  u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0, m, w;
  int i = 30;
  for (;;) {
    const int zeros = sg_ctz32(g | (~(u32)0 << i));
    g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
    if (i == 0) break;
    // f and g odd here
    if (eta < 0) {
      u32 tmp;
      eta = -eta;
      tmp = f; f = g; g = (u32)0 - tmp;
      tmp = u; u = q; q = (u32)0 - tmp;
      tmp = v; v = r; r = (u32)0 - tmp;
      const int limit = (eta + 1) > i ? i : (eta + 1);     // up to 6 steps at once: w = -g / f mod 2^limit
      m = (~(u32)0 >> (32 - limit)) & 63u;
      w = (f * g * (f * f - 2)) & m;
    } else {
      const int limit = (eta + 1) > i ? i : (eta + 1);     // up to 4 steps at once
      m = (~(u32)0 >> (32 - limit)) & 15u;
      w = f + (((f + 1) & 4) << 1);
      w = (((u32)0 - w) * g) & m;
    }
    g += f * w; q += u * w; r += v * w;
  }
  t->u = (i32)u; t->v = (i32)v; t->q = (i32)q; t->r = (i32)r;
  return eta;
}
// (f, g) <- t (f, g) / 2^30  (exact)
template <int NL> MZK_HD void sg_update_fg(Sgn30<NL>* f, Sgn30<NL>* g, const DivMat* t) {
  const i32 M30 = (i32)(~(u32)0 >> 2);
  const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
  int64_t cf = u * f->v[0] + v * g->v[0];
  int64_t cg = q * f->v[0] + r * g->v[0];
  cf >>= 30; cg >>= 30;
#pragma unroll
  for (int i = 1; i < NL; i++) {
    cf += u * f->v[i] + v * g->v[i];
    cg += q * f->v[i] + r * g->v[i];
    f->v[i - 1] = (i32)cf & M30; cf >>= 30;
    g->v[i - 1] = (i32)cg & M30; cg >>= 30;
  }
  f->v[NL - 1] = (i32)cf;
  g->v[NL - 1] = (i32)cg;
}
// (d, e) <- t (d, e) / 2^30 mod p, result again in (-2p, p)
template <int NL> MZK_HD void sg_update_de(Sgn30<NL>* d, Sgn30<NL>* e, const DivMat* t, const Sgn30<NL>* mod, u32 pinv30) {
  const i32 M30 = (i32)(~(u32)0 >> 2);
  const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
  const i32 sd = d->v[NL - 1] >> 31, se = e->v[NL - 1] >> 31;
  i32 md = (t->u & sd) + (t->v & se), me = (t->q & sd) + (t->r & se);
  int64_t cd = u * d->v[0] + v * e->v[0];
  int64_t ce = q * d->v[0] + r * e->v[0];
  md -= (i32)((pinv30 * (u32)cd + (u32)md) & (u32)M30);
  me -= (i32)((pinv30 * (u32)ce + (u32)me) & (u32)M30);
  cd += (int64_t)mod->v[0] * md;
  ce += (int64_t)mod->v[0] * me;
  cd >>= 30; ce >>= 30;
#pragma unroll
  for (int i = 1; i < NL; i++) {
    cd += u * d->v[i] + v * e->v[i] + (int64_t)mod->v[i] * md;
    ce += q * d->v[i] + r * e->v[i] + (int64_t)mod->v[i] * me;
    d->v[i - 1] = (i32)cd & M30; cd >>= 30;
    e->v[i - 1] = (i32)ce & M30; ce >>= 30;
  }
  d->v[NL - 1] = (i32)cd;
  e->v[NL - 1] = (i32)ce;
}
// r in (-2p, p), sign < 0 -> negate; result in [0, p)
template <int NL> MZK_HD void sg_normalize(Sgn30<NL>* r, i32 sign, const Sgn30<NL>* mod) {
  const i32 M30 = (i32)(~(u32)0 >> 2);
  i32 c[NL];
  i32 add = r->v[NL - 1] >> 31;
#pragma unroll
  for (int i = 0; i < NL; i++) c[i] = r->v[i] + (mod->v[i] & add);
  const i32 neg = sign >> 31;
#pragma unroll
  for (int i = 0; i < NL; i++) c[i] = (c[i] ^ neg) - neg;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) { c[i + 1] += c[i] >> 30; c[i] &= M30; }
  add = c[NL - 1] >> 31;
#pragma unroll
  for (int i = 0; i < NL; i++) c[i] += mod->v[i] & add;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) { c[i + 1] += c[i] >> 30; c[i] &= M30; }
#pragma unroll
  for (int i = 0; i < NL; i++) r->v[i] = c[i];
}
template <int NW, int NL> MZK_HD void sg_from_words(const u32* w, Sgn30<NL>* o) {   // NW u32 words -> 30-bit limbs
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = 30 * i, k = bit >> 5, sh = bit & 31;
    u64 x = (k < NW) ? w[k] : 0u;
    if (k + 1 < NW) x |= (u64)w[k + 1] << 32;
    o->v[i] = (i32)((u32)(x >> sh) & (~(u32)0 >> 2));
  }
}
template <int NW, int NL> MZK_HD void sg_to_words(const Sgn30<NL>* a, u32* w) {     // canonical value in [0, p) back to u32 words
#pragma unroll
  for (int k = 0; k < NW; k++) {
    // word k = bits [32k, 32k+32): from limb i = 32k / 30 (and the next one or two)
    const int bit = 32 * k, i = bit / 30, sh = bit - 30 * i;
    u64 x = (u64)(u32)a->v[i];
    if (i + 1 < NL) x |= (u64)(u32)a->v[i + 1] << 30;
    if (i + 2 < NL) x |= (u64)(u32)a->v[i + 2] << 60;
    w[k] = (u32)(x >> sh);
  }
}
template <class P> MZK_HD Fe<P> fe_inv_safegcd(const Fe<P>& a) {
  constexpr int NW = P::NW;
  constexpr int NL = (32 * NW + 29) / 30 + 0;      // 9 limbs for 256 bits, 5 for 128: the top limb carries the sign
  static_assert(30 * NL >= P::BITS + 2, "signed 30-bit limbs must hold values in (-2p, p)");
  u32 xw[NW], pw[NW];
  {
    Fe<P> c = fe_reduce<P>(a);
    if (fe_is_zero_canon<P>(c)) return fe_zero<P>();
    fe_pack<P>(c, xw);
  }
#pragma unroll
  for (int i = 0; i < NW; i++) pw[i] = P::PW[i];
  Sgn30<NL> mod, f, g, d, e;
  sg_from_words<NW, NL>(pw, &mod);
  sg_from_words<NW, NL>(xw, &g);
  f = mod;
#pragma unroll
  for (int i = 0; i < NL; i++) { d.v[i] = 0; e.v[i] = 0; }
  e.v[0] = 1;
  // p^-1 mod 2^30 by Newton from p0 (p odd: p0 * p0 == 1 mod 8)
  u32 pinv = (u32)mod.v[0];
#pragma unroll
  for (int it = 0; it < 4; it++) pinv *= 2 - (u32)mod.v[0] * pinv;
  pinv &= ~(u32)0 >> 2;
  i32 eta = -1;
  for (int batch = 0; batch < 32; batch++) {       // 25 batches bound 254-bit inputs; g == 0 ends it (~19-20)
    DivMat t;
    eta = sg_divsteps_30(eta, (u32)f.v[0], (u32)g.v[0], &t);
    sg_update_de<NL>(&d, &e, &t, &mod, pinv);
    sg_update_fg<NL>(&f, &g, &t);
    i32 gz = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) gz |= g.v[i];
    if (gz == 0) break;
  }
  sg_normalize<NL>(&d, f.v[NL - 1], &mod);           // f = +-1, d = +-x^-1
  u32 rw[NW];
  sg_to_words<NW, NL>(&d, rw);
  // rw = (aR)^-1 = a^-1 R^-1 (plain words); Montgomery form of a^-1 is a^-1 R = mont(rw, R^3)
  Fe<P> r3;
#pragma unroll
  for (int i = 0; i < P::L; i++) r3.l[i] = P::R3[i];
  return fe_mul<P>(fe_unpack<P>(rw), r3);
}

}  // namespace mzk
