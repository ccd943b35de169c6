// mzk_glv.h -- GLV decomposition of a BN254 scalar: k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^126.
//
// BN254 G1 has the endomorphism phi(x, y) = (beta x, y) = lambda (x, y) (beta^3 = 1 in Fq, lambda^2 + lambda + 1 = 0
// in Fr), so  k P = k1 P + k2 phi(P).  The bucket method then runs on 2n points with half-length scalars: the same
// number of bucket additions, but half the bucket sets to reduce and -- what matters on a GPU -- a window Horner of
// 112 instead of 240 serial doublings.  Group-theoretic identity, so results stay bit-identical to the reference's
// double-and-add (polynomial.rs:156-165) for points ON the curve.
//
// With the short lattice basis v1 = (A1, -B1N), v2 = (A2, B2) of {(a, b): a + b lambda = 0 mod r}:
//   c1 = round(B2 k / r), c2 = round(B1N k / r),   k1 = k - c1 A1 - c2 A2,   k2 = c1 B1N - c2 B2.
// k1 + k2 lambda = k (mod r) holds for ANY integers c1, c2 (v1, v2 are lattice vectors): the rounding only controls
// the size.  round(x k / r) is taken as (k G + 2^319) >> 320 with G = round(x 2^320 / r): off by one only within
// 2^-60 of a rounding boundary, where both neighbours leave a remainder of half a basis vector, so
// |k1| <= (A1 + A2) / 2 (1 + eps) < 2^126 and likewise |k2|.  Plain C++: runs under the host checker.
#pragma once
#include "mzk_field.h"

namespace mzk {

// r[0..na+nb) = a * b (32-bit words, little endian)
template <int NA, int NB> MZK_HD void w_mul(const u32* a, const u32* b, u32* r) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      c += (u64)a[i] * b[j] + r[i + j];
      r[i + j] = (u32)c;
      c >>= 32;
    }
    r[i + NB] = (u32)c;
  }
}
// (k * g + 2^319) >> 320 for an 8-word k and an NG-word g; result in NR words
template <int NG, int NR> MZK_HD void glv_round_quot(const u32* k, const u32* g, u32* out) {
  u32 prod[8 + NG + 1];
  w_mul<8, NG>(k, g, prod);
  prod[8 + NG] = 0;
  // add 2^319 = bit 31 of word 9
  u64 c = (u64)prod[9] + 0x80000000u;
  prod[9] = (u32)c;
  c >>= 32;
#pragma unroll
  for (int i = 10; i < 8 + NG + 1; i++) { c += prod[i]; prod[i] = (u32)c; c >>= 32; }
#pragma unroll
  for (int i = 0; i < NR; i++) out[i] = (10 + i < 8 + NG + 1) ? prod[10 + i] : 0u;
}
// two's-complement 8-word helpers
MZK_HD void w8_sub(u32* a, const u32* b) {   // a -= b (mod 2^256)
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { const u64 d = (u64)a[i] - b[i] - br; a[i] = (u32)d; br = (d >> 63) & 1; }
}
MZK_HD void w8_sign_mag(const u32* v, u32* mag4, u32* neg) {   // |v| < 2^127 assumed: magnitude in 4 words
  const u32 s = v[7] >> 31;
  u32 t[8];
  u64 c = s;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (u64)(s ? ~v[i] : v[i]); t[i] = (u32)c; c >>= 32; }
#pragma unroll
  for (int i = 0; i < 4; i++) mag4[i] = t[i];
  MZK_ASSERT((t[4] | t[5] | t[6] | t[7]) == 0 && (t[3] >> 30) == 0);   // < 2^126
  *neg = s;
}
// k: canonical scalar (8 words).  m1, m2: magnitudes (4 words, < 2^126); neg1, neg2: 1 if the part is negative.
MZK_HD void glv_split(const u32* k, u32* m1, u32* neg1, u32* m2, u32* neg2) {
  u32 g1[5], g2[7], a1[2], a2[4], b1n[4], b2[2];
#pragma unroll
  for (int i = 0; i < 5; i++) g1[i] = GlvParams::G1[i];
#pragma unroll
  for (int i = 0; i < 7; i++) g2[i] = GlvParams::G2[i];
#pragma unroll
  for (int i = 0; i < 2; i++) { a1[i] = GlvParams::A1[i]; b2[i] = GlvParams::B2[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) { a2[i] = GlvParams::A2[i]; b1n[i] = GlvParams::B1N[i]; }
  u32 c1[3], c2[4];
  glv_round_quot<5, 3>(k, g1, c1);     // < 2^65
  glv_round_quot<7, 4>(k, g2, c2);     // < 2^128
  u32 t5[5], t8[8], t7[7], t6[6], v[8];
  // k1 = k - c1 A1 - c2 A2
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = k[i];
  w_mul<3, 2>(c1, a1, t5);
  { u32 e[8] = {t5[0], t5[1], t5[2], t5[3], t5[4], 0, 0, 0}; w8_sub(v, e); }
  w_mul<4, 4>(c2, a2, t8);
  w8_sub(v, t8);
  w8_sign_mag(v, m1, neg1);
  // k2 = c1 B1N - c2 B2
  w_mul<3, 4>(c1, b1n, t7);
  { u32 e[8] = {t7[0], t7[1], t7[2], t7[3], t7[4], t7[5], t7[6], 0};
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = e[i]; }
  w_mul<4, 2>(c2, b2, t6);
  { u32 e[8] = {t6[0], t6[1], t6[2], t6[3], t6[4], t6[5], 0, 0}; w8_sub(v, e); }
  w8_sign_mag(v, m2, neg2);
}

}  // namespace mzk
