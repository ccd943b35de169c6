// mzk_g2.h -- BN254 G2: Fq2 = Fq[u]/(u^2 + 1) on the 29-bit-limb Montgomery field code, XYZZ group law over it.
//
// Reference: Fq2 / G2Point bn128.rs:33-49 (ExtendedFieldElement over x^2 + 1, efield.rs), group law curve.rs:56-191,
// MSM call site kzg.rs:114, powers_2 of setup_kzg_with_full_g2 kzg.rs:42-55.  G2 work in the reference is small
// (k+1-term MSMs in batch verification, max_d+1 fixed-base multiples in the full-G2 setup), so this code favours
// simple invariants over the last cycle: EVERY Fq value held in an Fq2 is "N2" -- normalised limbs, value < 2.01 p
// (what fe_weak_reduce and fe_mul return) -- and every operation re-establishes that.  Plain C++ so that the
// bounds-checked host build (tests/hostcheck) can run it.
#pragma once
#include "mzk_ec.h"

namespace mzk {

struct Fq2 { Fq c0, c1; };   // c0 + c1 u
typedef FqParams QP;

MZK_HD Fq2 f2_zero() { Fq2 r; r.c0 = fe_zero<QP>(); r.c1 = fe_zero<QP>(); return r; }
MZK_HD Fq2 f2_one() { Fq2 r; r.c0 = fe_one<QP>(); r.c1 = fe_zero<QP>(); return r; }
MZK_HD Fq2 f2_add(const Fq2& a, const Fq2& b) {
  Fq2 r; r.c0 = fe_weak_reduce<QP>(fe_add<QP>(a.c0, b.c0)); r.c1 = fe_weak_reduce<QP>(fe_add<QP>(a.c1, b.c1)); return r;
}
MZK_HD Fq2 f2_sub(const Fq2& a, const Fq2& b) {
  Fq2 r; r.c0 = fe_weak_reduce<QP>(fe_sub<QP, 8>(a.c0, b.c0)); r.c1 = fe_weak_reduce<QP>(fe_sub<QP, 8>(a.c1, b.c1)); return r;
}
MZK_HD Fq2 f2_dbl(const Fq2& a) { return f2_add(a, a); }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u: four products, two Montgomery reductions
MZK_HD Fq2 f2_mul(const Fq2& a, const Fq2& b) {
  Fq2 r;
  r.c0 = fe_mul_add2<QP>(a.c0, b.c0, fe_neg_lazy<QP, 8>(a.c1), b.c1);
  r.c1 = fe_mul_add2<QP>(a.c0, b.c1, a.c1, b.c0);
  return r;
}
MZK_HD Fq2 f2_sqr(const Fq2& a) { return f2_mul(a, a); }
MZK_HD bool f2_is_zero(const Fq2& a) { return fe_is_zero_mod<QP, 3>(a.c0) && fe_is_zero_mod<QP, 3>(a.c1); }
MZK_HD Fq2 f2_reduce(const Fq2& a) { Fq2 r; r.c0 = fe_reduce<QP>(a.c0); r.c1 = fe_reduce<QP>(a.c1); return r; }
// conj(a) / (a0^2 + a1^2); 0 -> 0 like ExtendedFieldElement::inverse (efield.rs:126-151)
MZK_HD Fq2 f2_inv(const Fq2& a) {
  const Fq n = fe_mul_add2<QP>(a.c0, a.c0, a.c1, a.c1);
  const Fq ni = fe_inv<QP>(n);
  Fq2 r;
  r.c0 = fe_mul<QP>(a.c0, ni);
  r.c1 = fe_mul<QP>(fe_neg_lazy<QP, 8>(a.c1), ni);
  return r;
}
// wire <-> registers: 8 canonical words per coefficient
MZK_HD Fq2 f2_load_plain(const u32* w) { Fq2 r; r.c0 = fe_to_mont<QP>(fe_unpack<QP>(w)); r.c1 = fe_to_mont<QP>(fe_unpack<QP>(w + 8)); return r; }
MZK_HD void f2_store_plain(const Fq2& a, u32* w) { fe_pack<QP>(fe_from_mont<QP>(a.c0), w); fe_pack<QP>(fe_from_mont<QP>(a.c1), w + 8); }
MZK_HD Fq2 f2_load_raw(const u32* w) { Fq2 r; r.c0 = fe_unpack<QP>(w); r.c1 = fe_unpack<QP>(w + 8); return r; }   // Montgomery words
MZK_HD void f2_store_raw(const Fq2& a, u32* w) { fe_pack<QP>(fe_reduce<QP>(a.c0), w); fe_pack<QP>(fe_reduce<QP>(a.c1), w + 8); }

struct Affine2 { Fq2 x, y; };
struct Xyzz2 { Fq2 X, Y, ZZ, ZZZ; };     // x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ = 0 (all-zero limbs)
MZK_HD Xyzz2 x2_inf() { Xyzz2 r; r.X = f2_zero(); r.Y = f2_zero(); r.ZZ = f2_zero(); r.ZZZ = f2_zero(); return r; }
MZK_HD bool x2_is_inf(const Xyzz2& p) {
  u32 acc = 0;
  for (int i = 0; i < QP::L; i++) acc |= p.ZZ.c0.l[i] | p.ZZ.c1.l[i];
  return acc == 0;
}
MZK_HD Xyzz2 x2_from_affine(const Affine2& a) { Xyzz2 r; r.X = a.x; r.Y = a.y; r.ZZ = f2_one(); r.ZZZ = f2_one(); return r; }
// dbl-2008-s-1 (a = 0)
MZK_HD Xyzz2 x2_dbl(const Xyzz2& p) {
  if (x2_is_inf(p)) return p;
  const Fq2 U = f2_dbl(p.Y);
  if (f2_is_zero(U)) return x2_inf();                  // 2-torsion cannot occur in the prime-order group; kept for completeness
  const Fq2 V = f2_sqr(U), W = f2_mul(U, V), S = f2_mul(p.X, V);
  const Fq2 X2 = f2_sqr(p.X);
  const Fq2 M = f2_add(f2_dbl(X2), X2);
  Xyzz2 r;
  r.X = f2_sub(f2_sub(f2_sqr(M), S), S);
  r.Y = f2_sub(f2_mul(M, f2_sub(S, r.X)), f2_mul(W, p.Y));
  r.ZZ = f2_mul(V, p.ZZ);
  r.ZZZ = f2_mul(W, p.ZZZ);
  return r;
}
MZK_HD Xyzz2 x2_dbl_affine(const Affine2& a) { return x2_dbl(x2_from_affine(a)); }
// madd-2008-s, exception-complete (curve.rs:104-115: same point -> double, opposite -> infinity)
MZK_HD Xyzz2 x2_madd(const Xyzz2& a, const Affine2& q) {
  if (x2_is_inf(a)) return x2_from_affine(q);
  const Fq2 U2 = f2_mul(q.x, a.ZZ), S2 = f2_mul(q.y, a.ZZZ);
  const Fq2 P = f2_sub(U2, a.X), R = f2_sub(S2, a.Y);
  if (f2_is_zero(P)) {
    if (f2_is_zero(R)) return x2_dbl_affine(q);
    return x2_inf();
  }
  const Fq2 PP = f2_sqr(P), PPP = f2_mul(P, PP), Q = f2_mul(a.X, PP);
  Xyzz2 r;
  r.X = f2_sub(f2_sub(f2_sub(f2_sqr(R), PPP), Q), Q);
  r.Y = f2_sub(f2_mul(R, f2_sub(Q, r.X)), f2_mul(a.Y, PPP));
  r.ZZ = f2_mul(a.ZZ, PP);
  r.ZZZ = f2_mul(a.ZZZ, PPP);
  return r;
}
// add-2008-s, exception-complete
MZK_HD Xyzz2 x2_add(const Xyzz2& a, const Xyzz2& b) {
  if (x2_is_inf(a)) return b;
  if (x2_is_inf(b)) return a;
  const Fq2 U1 = f2_mul(a.X, b.ZZ), U2 = f2_mul(b.X, a.ZZ), S1 = f2_mul(a.Y, b.ZZZ), S2 = f2_mul(b.Y, a.ZZZ);
  const Fq2 P = f2_sub(U2, U1), R = f2_sub(S2, S1);
  if (f2_is_zero(P)) {
    if (f2_is_zero(R)) return x2_dbl(a);
    return x2_inf();
  }
  const Fq2 PP = f2_sqr(P), PPP = f2_mul(P, PP), Q = f2_mul(U1, PP);
  Xyzz2 r;
  r.X = f2_sub(f2_sub(f2_sub(f2_sqr(R), PPP), Q), Q);
  r.Y = f2_sub(f2_mul(R, f2_sub(Q, r.X)), f2_mul(S1, PPP));
  r.ZZ = f2_mul(f2_mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = f2_mul(f2_mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}
MZK_HD bool x2_to_affine(const Xyzz2& p, Affine2* out) {
  if (x2_is_inf(p)) return false;
  const Fq2 di = f2_inv(f2_mul(p.ZZ, p.ZZZ));
  out->x = f2_mul(p.X, f2_mul(di, p.ZZZ));
  out->y = f2_mul(p.Y, f2_mul(di, p.ZZ));
  return true;
}
// k * q for a canonical 8-word scalar, MSB-first double-and-add on the XYZZ accumulator
MZK_HD Xyzz2 x2_scalar_mul(const Affine2& q, const u32* k) {
  Xyzz2 acc = x2_inf();
  bool started = false;
  for (int bit = 255; bit >= 0; bit--) {
    if (started) acc = x2_dbl(acc);
    if ((k[bit >> 5] >> (bit & 31)) & 1u) { acc = x2_madd(acc, q); started = true; }
  }
  return acc;
}
// wire point: x.c0 | x.c1 | y.c0 | y.c1, 8 canonical words each; all-zero = infinity
MZK_HD bool g2_words_is_inf(const u32* w) { u32 acc = 0; for (int i = 0; i < 32; i++) acc |= w[i]; return acc == 0; }
MZK_HD Affine2 g2_load_plain(const u32* w) { Affine2 a; a.x = f2_load_plain(w); a.y = f2_load_plain(w + 16); return a; }
MZK_HD void g2_store_plain(const Xyzz2& p, u32* w) {
  Affine2 a;
  if (!x2_to_affine(p, &a)) { for (int i = 0; i < 32; i++) w[i] = 0; return; }
  f2_store_plain(a.x, w); f2_store_plain(a.y, w + 16);
}
// XYZZ record in memory: 8 coefficients x 8 Montgomery words (64 words); all-zero = infinity
MZK_HD void x2_store(const Xyzz2& p, u32* w) {
  if (x2_is_inf(p)) { for (int i = 0; i < 64; i++) w[i] = 0; return; }
  f2_store_raw(p.X, w); f2_store_raw(p.Y, w + 16); f2_store_raw(p.ZZ, w + 32); f2_store_raw(p.ZZZ, w + 48);
}
MZK_HD Xyzz2 x2_load(const u32* w) { Xyzz2 p; p.X = f2_load_raw(w); p.Y = f2_load_raw(w + 16); p.ZZ = f2_load_raw(w + 32); p.ZZZ = f2_load_raw(w + 48); return p; }

}  // namespace mzk
