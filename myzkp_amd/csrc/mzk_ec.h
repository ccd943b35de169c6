// mzk_ec.h -- BN254 G1 group law for the MSM kernels (y^2 = x^3 + 3 over Fq, a = 0).
//
// The reference adds points in affine coordinates with one field inversion per operation
// (EllipticCurvePoint::add_ref / double, myzkp/src/modules/algebra/curve/curve.rs:72-161).  The
// kernels use inversion-free XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) and convert once at
// the end; the group element -- hence the canonical affine output -- is identical.  Every special
// case the reference distinguishes is kept explicit (curve.rs:104-115): inf + Q, P + inf,
// P + P -> double, P + (-P) -> inf, so adversarial inputs (repeated points, P next to -P) agree.
//
// Bounds bookkeeping (units of p; "N" = normalised limbs, see mzk_field.h): coordinates of a stored
// accumulator are N and < 2.5; fe_mul/fe_sqr outputs are N and < 2; fe_weak_reduce outputs are N and
// < 2.01.  With rho = p/R ~ 2^-7.4 a product of inputs (va, vb) is < va*vb*rho + 1.
#pragma once
#include "mzk_field.h"
#include "mzk_field_asm.h"

namespace mzk {

typedef Fe<FqParams> Fq;

struct Affine { Fq x, y; };            // Montgomery form, canonical; never infinity in registers
struct Xyzz { Fq X, Y, ZZ, ZZZ; };     // infinity <=> ZZ limbs all zero (written explicitly)

MZK_HD bool xyzz_is_inf(const Xyzz& p) { return fe_is_zero_canon<FqParams>(p.ZZ); }
MZK_HD Xyzz xyzz_inf() {
  Xyzz r;
  r.X = fe_zero<FqParams>(); r.Y = fe_zero<FqParams>(); r.ZZ = fe_zero<FqParams>(); r.ZZZ = fe_zero<FqParams>();
  return r;
}
MZK_HD Xyzz xyzz_from_affine(const Affine& a) {
  Xyzz r;
  r.X = a.x; r.Y = a.y; r.ZZ = fe_one<FqParams>(); r.ZZZ = fe_one<FqParams>();
  return r;
}
// -(x, y) = (x, p - y); y canonical and non-zero on this curve (no 2-torsion: the group order r is odd)
MZK_HD Affine affine_neg(const Affine& a) {
  Affine r;
  r.x = a.x;
  r.y = fe_neg_canon<FqParams>(a.y);
  return r;
}

// 2 * (affine) -> XYZZ   (mdbl-2008-s-1, a = 0): 4M + 3S
MZK_HD Xyzz xyzz_dbl_affine(const Affine& a) {
  typedef FqParams P;
  Xyzz r;
  Fq U = fe_dbl<P>(a.y);                         // < 2, limbs < 2^30
  Fq V = fe_sqr<P>(U);                           // < 1.03
  Fq W = fe_mul<P>(U, V);                        // < 1.02
  Fq S = fe_mul<P>(a.x, V);                      // < 1.01
  Fq X2 = fe_sqr<P>(a.x);                        // < 1.01
  Fq M = fe_carry<P>(fe_add<P>(fe_dbl<P>(X2), X2));  // 3 x^2 < 3.03, N
  Fq MM = fe_sqr<P>(M);                          // < 1.06
  Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(MM, S), S));  // MM - 2S (+8p) -> < 2.01
  Fq Vd = fe_carry<P>(fe_sub<P, 8>(S, X3));      // S - X3 (+8p) < 9.01, N
  r.X = X3;
  r.Y = fe_mul_add2<P>(M, Vd, fe_neg_lazy<P, 4>(a.y), W);      // < 1.2
  r.ZZ = V;
  r.ZZZ = W;
  return r;
}

// Product routines of the group law: FeCpp (portable C++, scheduled by the compiler: best for the latency-bound tail
// kernels) or FeAsm (mzk_field_asm.h: one inline-asm chain per product, fewer instructions: best at 3+ waves per SIMD).
// Both compute the same column sums, so the result limbs are identical.
template <class P> struct FeCpp {
  static MZK_HD Fe<P> mul(const Fe<P>& a, const Fe<P>& b) { return fe_mul<P>(a, b); }
  static MZK_HD Fe<P> sqr(const Fe<P>& a) { return fe_sqr<P>(a); }
  static MZK_HD Fe<P> mul_add2(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) { return fe_mul_add2<P>(a, b, c, d); }
};

// 2 * (XYZZ) -> XYZZ   (dbl-2008-s-1, a = 0): 6M + 3S.  inf -> inf.
template <template <class> class A> MZK_HD Xyzz xyzz_dbl_with(const Xyzz& p) {
  typedef FqParams P;
  typedef A<P> F;
  if (xyzz_is_inf(p)) return p;
  Xyzz r;
  Fq U = fe_dbl<P>(p.Y);                         // < 5, limbs < 2^30
  Fq V = F::sqr(U);                              // < 1.15
  Fq W = F::mul(U, V);                           // < 1.04
  Fq S = F::mul(p.X, V);                         // < 1.02
  Fq X2 = F::sqr(p.X);                           // < 1.04
  Fq M = fe_carry<P>(fe_add<P>(fe_dbl<P>(X2), X2));  // < 3.12, N
  Fq MM = F::sqr(M);                             // < 1.06
  Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(MM, S), S));
  Fq Vd = fe_carry<P>(fe_sub<P, 8>(S, X3));      // < 9.02
  r.X = X3;
  r.Y = F::mul_add2(M, Vd, fe_neg_lazy<P, 8>(p.Y), W);         // M (S - X3) - Y1 W: < 1.22
  r.ZZ = F::mul(V, p.ZZ);
  r.ZZZ = F::mul(W, p.ZZZ);
  return r;
}
MZK_HD Xyzz xyzz_dbl(const Xyzz& p) { return xyzz_dbl_with<FeCpp>(p); }

// acc + (affine q) -> XYZZ   (madd-2008-s): 8M + 2S.  Exception-complete.
// A: FeCpp or FeAsm (above).
template <template <class> class A> MZK_HD Xyzz xyzz_madd_with(const Xyzz& a, const Affine& q) {
  typedef FqParams P;
  typedef A<P> F;
  if (xyzz_is_inf(a)) return xyzz_from_affine(q);
  Fq U2 = F::mul(q.x, a.ZZ);                     // < 1.02
  Fq S2 = F::mul(q.y, a.ZZZ);                    // < 1.02
  Fq Pd = fe_carry<P>(fe_sub<P, 8>(U2, a.X));    // U2 - X1 (+8p) < 9.02, N
  Fq Rd = fe_carry<P>(fe_sub<P, 8>(S2, a.Y));    // < 9.02, N
  if (fe_is_zero_mod<P, 10>(Pd)) {               // same x: q == +-a   (curve.rs:111-115)
    if (fe_is_zero_mod<P, 10>(Rd)) return xyzz_dbl_affine(q);
    return xyzz_inf();
  }
  Xyzz r;
  Fq PP = F::sqr(Pd);                            // < 1.49
  Fq PPP = F::mul(Pd, PP);                       // < 1.08
  Fq Q = F::mul(a.X, PP);                        // < 1.03
  Fq RR = F::sqr(Rd);                            // < 1.49
  Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(fe_sub<P, 4>(RR, PPP), Q), Q));  // (+12p) < 13.5 -> < 2.01
  Fq Vd = fe_carry<P>(fe_sub<P, 8>(Q, X3));      // < 9.03
  r.X = X3;
  // Y3 = Rd (Q - X3) - Y1 PPP as one fused product pair: Rd Vd + (8p - Y1) PPP, single reduction
  r.Y = F::mul_add2(Rd, Vd, fe_neg_lazy<P, 8>(a.Y), PPP);      // < (9.02*9.03 + 8*1.08) rho + 1 < 1.54
  r.ZZ = F::mul(a.ZZ, PP);
  r.ZZZ = F::mul(a.ZZZ, PPP);
  return r;
}
MZK_HD Xyzz xyzz_madd(const Xyzz& a, const Affine& q) { return xyzz_madd_with<FeCpp>(a, q); }

// acc + (neg ? -q : q): the signed-digit entries of the bucket method.  The sign is applied to S2 = y2 ZZZ1 after the product
// (limb-wise 4p - S2 and a select: 18 instructions) instead of to y2 before it (canonical p - y: a carry chain, a zero test
// and two selects: ~55); y2 enters the formula nowhere else.  Same exception handling as xyzz_madd_with.
template <template <class> class A> MZK_HD Xyzz xyzz_madd_signed_with(const Xyzz& a, const Affine& q, bool neg) {
  typedef FqParams P;
  typedef A<P> F;
  if (xyzz_is_inf(a)) return xyzz_from_affine(neg ? affine_neg(q) : q);
  Fq U2 = F::mul(q.x, a.ZZ);                     // < 1.02
  Fq S2p = F::mul(q.y, a.ZZZ);                   // < 1.02
  Fq S2n = fe_neg_lazy<P, 4>(S2p);               // 4p - S2 < 4, limbs < 2^30.6
  Fq S2;
#pragma unroll
  for (int i = 0; i < P::L; i++) { const u32 vp = S2p.l[i], vn = S2n.l[i]; S2.l[i] = neg ? vn : vp; }
  Fq Pd = fe_carry<P>(fe_sub<P, 8>(U2, a.X));    // U2 - X1 (+8p) < 9.02, N
  Fq Rd = fe_carry<P>(fe_sub<P, 8>(S2, a.Y));    // +-S2 - Y1 (+8p or +12p) < 12, N
  if (fe_is_zero_mod<P, 10>(Pd)) {               // same x: +-q == +-a   (curve.rs:111-115)
    if (fe_is_zero_mod<P, 12>(Rd)) return xyzz_dbl_affine(neg ? affine_neg(q) : q);
    return xyzz_inf();
  }
  Xyzz r;
  Fq PP = F::sqr(Pd);                            // < 1.49
  Fq PPP = F::mul(Pd, PP);                       // < 1.08
  Fq Q = F::mul(a.X, PP);                        // < 1.03
  Fq RR = F::sqr(Rd);                            // < 1.86
  Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(fe_sub<P, 4>(RR, PPP), Q), Q));  // (+12p) < 13.9 -> < 2.01
  Fq Vd = fe_carry<P>(fe_sub<P, 8>(Q, X3));      // < 9.03
  r.X = X3;
  r.Y = F::mul_add2(Rd, Vd, fe_neg_lazy<P, 8>(a.Y), PPP);      // < (12*9.03 + 8*1.08) rho + 1 < 1.7
  r.ZZ = F::mul(a.ZZ, PP);
  r.ZZZ = F::mul(a.ZZZ, PPP);
  return r;
}

// a + b, both XYZZ   (add-2008-s): 12M + 2S.  Exception-complete.  A as in xyzz_madd_with.
template <template <class> class A> MZK_HD Xyzz xyzz_add_with(const Xyzz& a, const Xyzz& b) {
  typedef FqParams P;
  typedef A<P> F;
  if (xyzz_is_inf(a)) return b;
  if (xyzz_is_inf(b)) return a;
  Fq U1 = F::mul(a.X, b.ZZ);                     // < 1.04
  Fq U2 = F::mul(b.X, a.ZZ);
  Fq S1 = F::mul(a.Y, b.ZZZ);
  Fq S2 = F::mul(b.Y, a.ZZZ);
  Fq Pd = fe_carry<P>(fe_sub<P, 4>(U2, U1));     // < 5.04, N
  Fq Rd = fe_carry<P>(fe_sub<P, 4>(S2, S1));
  if (fe_is_zero_mod<P, 6>(Pd)) {
    if (fe_is_zero_mod<P, 6>(Rd)) return xyzz_dbl(a);
    return xyzz_inf();
  }
  Xyzz r;
  Fq PP = F::sqr(Pd);                            // < 1.15
  Fq PPP = F::mul(Pd, PP);                       // < 1.04
  Fq Q = F::mul(U1, PP);                         // < 1.01
  Fq RR = F::sqr(Rd);                            // < 1.15
  Fq X3 = fe_weak_reduce<P>(fe_sub<P, 4>(fe_sub<P, 4>(fe_sub<P, 4>(RR, PPP), Q), Q));
  Fq Vd = fe_carry<P>(fe_sub<P, 8>(Q, X3));      // < 9.02
  r.X = X3;
  r.Y = F::mul_add2(Rd, Vd, fe_neg_lazy<P, 4>(S1), PPP);       // Rd Vd + (4p - S1) PPP: < 1.3
  r.ZZ = F::mul(F::mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = F::mul(F::mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}
MZK_HD Xyzz xyzz_add(const Xyzz& a, const Xyzz& b) { return xyzz_add_with<FeCpp>(a, b); }

// XYZZ -> canonical affine (Montgomery form); returns false for infinity.  SERIAL selects the
// single-lane inversion (tail kernels); the default is the divergence-free Fermat ladder.
template <bool SERIAL = false> MZK_HD bool xyzz_to_affine(const Xyzz& p, Affine* out) {
  typedef FqParams P;
  if (xyzz_is_inf(p)) return false;
  Fq d = fe_mul<P>(p.ZZ, p.ZZZ);
  Fq di = SERIAL ? fe_inv_safegcd<P>(d) : fe_inv<P>(d);   // 1 / (ZZ ZZZ)
  Fq izz = fe_mul<P>(di, p.ZZZ);                 // 1 / ZZ
  Fq izzz = fe_mul<P>(di, p.ZZ);                 // 1 / ZZZ
  out->x = fe_reduce<P>(fe_mul<P>(p.X, izz));
  out->y = fe_reduce<P>(fe_mul<P>(p.Y, izzz));
  return true;
}

// ---- ABI <-> registers ------------------------------------------------------------------------
// Affine point at the ABI: 16 u32 words x||y canonical, all-zero = infinity.
MZK_HD bool affine_words_is_inf(const u32* w) {
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) acc |= w[i];
  return acc == 0;
}
// canonical plain words -> Montgomery registers
MZK_HD Affine affine_load_plain(const u32* w) {
  Affine a;
  a.x = fe_reduce<FqParams>(fe_to_mont<FqParams>(fe_unpack<FqParams>(w)));
  a.y = fe_reduce<FqParams>(fe_to_mont<FqParams>(fe_unpack<FqParams>(w + 8)));
  return a;
}
// Montgomery (canonical) words, as written by the point-preparation kernel
MZK_HD Affine affine_load_mont(const u32* w) {
  Affine a;
  a.x = fe_unpack<FqParams>(w);
  a.y = fe_unpack<FqParams>(w + 8);
  return a;
}
MZK_HD void affine_store_mont(const Affine& a, u32* w) {
  fe_pack<FqParams>(a.x, w);
  fe_pack<FqParams>(a.y, w + 8);
}
MZK_HD void affine_store_plain(const Affine& a, u32* w) {
  fe_pack<FqParams>(fe_from_mont<FqParams>(a.x), w);
  fe_pack<FqParams>(fe_from_mont<FqParams>(a.y), w + 8);
}
// XYZZ in global memory: 4 x 8 words, each coordinate fully reduced (fits 256 bits); inf = zeros.
MZK_HD void xyzz_store(const Xyzz& p, u32* w) {
  if (xyzz_is_inf(p)) {
#pragma unroll
    for (int i = 0; i < 32; i++) w[i] = 0;
    return;
  }
  fe_pack<FqParams>(fe_reduce<FqParams>(p.X), w);
  fe_pack<FqParams>(fe_reduce<FqParams>(p.Y), w + 8);
  fe_pack<FqParams>(fe_reduce<FqParams>(p.ZZ), w + 16);
  fe_pack<FqParams>(fe_reduce<FqParams>(p.ZZZ), w + 24);
}
MZK_HD Xyzz xyzz_load(const u32* w) {
  Xyzz p;
  p.X = fe_unpack<FqParams>(w);
  p.Y = fe_unpack<FqParams>(w + 8);
  p.ZZ = fe_unpack<FqParams>(w + 16);
  p.ZZZ = fe_unpack<FqParams>(w + 24);
  return p;
}

}  // namespace mzk
