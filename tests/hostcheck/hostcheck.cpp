// Host build of the device arithmetic headers with bounds assertions (-DMZK_CHECK_BOUNDS).
// TEST INFRASTRUCTURE: lets the CPU test-suite check the exact limb code the HIP kernels run
// (29-bit-limb Montgomery field ops, XYZZ group law) against the oracle.  Never shipped.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../myzkp_amd/csrc/mzk_ec.h"
#include "../../myzkp_amd/csrc/mzk_g2.h"
#include "../../myzkp_amd/csrc/mzk_glv.h"
using namespace mzk;

template <class P> static void run_field(int op, const u32* a, const u32* b, u32* out) {
  Fe<P> x = fe_to_mont<P>(fe_unpack<P>(a));
  Fe<P> y = fe_to_mont<P>(fe_unpack<P>(b));
  Fe<P> r;
  switch (op) {
    case 0: r = fe_mul<P>(x, y); break;
    case 1: r = fe_sqr<P>(x); break;
    case 2: r = fe_add<P>(x, y); break;
    case 3: r = fe_sub<P, 4>(x, fe_reduce<P>(y)); break;
    case 4: r = fe_inv_serial<P>(x); break;
    case 8: r = fe_inv<P>(x); break;
    case 10: r = fe_inv_safegcd<P>(x); break;
    case 11: r = fe_inv_safegcd<P>(fe_add<P>(fe_add<P>(x, x), fe_sub<P, 4>(x, fe_reduce<P>(x)))); break;
    case 9: r = fe_inv_serial<P>(fe_add<P>(fe_add<P>(x, x), fe_sub<P, 4>(x, fe_reduce<P>(x)))); break;  // inverse of a lazy 2x (+4p)
    case 5: r = fe_neg_canon<P>(fe_reduce<P>(x)); break;
    case 6: {  // lazy chain stressing fe_weak_reduce: ((x+y)+(x+y)) + 16p - y ... then squared
      Fe<P> s = fe_add<P>(x, y);
      s = fe_carry<P>(fe_add<P>(s, s));
      s = fe_sub<P, 16>(s, fe_reduce<P>(y));
      r = fe_sqr<P>(fe_weak_reduce<P>(s));
      break;
    }
    case 7: {  // plain-domain product by a Montgomery constant (the NTT trick): unpack(a) * mont(b)
      r = fe_mul<P>(fe_unpack<P>(a), y);
      fe_pack<P>(fe_reduce<P>(r), out);
      return;
    }
    default: r = fe_zero<P>();
  }
  fe_pack<P>(fe_from_mont<P>(r), out);
}

// The in-tile stage schedule of mzk_ntt.hip (tile_stages / bfly) on a plain array, with the bounds assertions on: radix-2 DIT
// over bit-reversed input, stages in pairs -- the FIRST stage of a pair keeps its sums limb-wise (fe_add / fe_sub<8>, no carry
// propagation), the second normalises (fe_add_carry / fe_sub_carry<8>); a twiddle of 1 skips the product (weak reduction
// instead, except on raw stage-1 inputs); an odd level count starts with one carrying stage.  Data plain, twiddles
// Montgomery, like the kernels.  words: n x NW in, n x NW out (reduced, natural order); tw: n/2 twiddles w^j, plain words.
// shoup (Fr only, may be null): 18 words per twiddle -- the 29-bit limbs of the plain w^j and of floor(w^j 2^261 / p); stage
// pairs in which a whole wave of the kernel shares a twiddle (2^(lgn + lgc - 2 - (s - 1)) >= 64 groups per j1; lgc = log2 of the
// tile's columns) then multiply by fe_shoup_mul instead of the Montgomery product, as tile_stages / radix4_shoup do.
template <class P> static void run_ntt_tile(const u32* words, int lgn, const u32* tw_words, u32* out, const u32* shoup = nullptr, int lgc = 0) {
  const int n = 1 << lgn;
  std::vector<Fe<P>> x(n), tw(n / 2 > 0 ? n / 2 : 1);
  for (int j = 0; j < n / 2; j++) tw[j] = fe_reduce<P>(fe_to_mont<P>(fe_unpack<P>(tw_words + (size_t)j * P::NW)));
  for (int j = 0; j < n; j++) {
    int k = 0;
    for (int b = 0; b < lgn; b++) if (j & (1 << b)) k |= 1 << (lgn - 1 - b);
    x[k] = fe_unpack<P>(words + (size_t)j * P::NW);
  }
  bool use_shoup = false;
  if constexpr (SparseMod<P>::value) {
    // signed lazy schedule of the sparse-modulus fields (mzk_ntt.hip: sbfly / radix4_regs): limb-wise sums of i32 limbs, the
    // unmultiplied inputs of a stage pair carried on entry, signed products, fe_sreduce at the end; every bound asserted
    auto sb = [&](Fe<P>& lo, Fe<P>& hi, int ti, bool trivial) {
      Fe<P> t = hi;
      if (!trivial) t = fe_mul_sparse<P, true, 0>(t, tw[ti]);
      hi = fe_ssub<P>(lo, t);
      lo = fe_sadd<P>(lo, t);
    };
    int s = 1;
    if (lgn & 1) {
      for (int g = 0; g < n / 2; g++) sb(x[2 * g], x[2 * g + 1], 0, true);
      s = 2;
    }
    for (; s + 1 <= lgn; s += 2) {
      const int lgh = s - 1, half = 1 << lgh;
      for (int grp = 0; grp < (n >> (s + 1)); grp++)
        for (int j1 = 0; j1 < half; j1++) {
          const int p0 = (grp << (s + 1)) | j1, d1 = half, d2 = half << 1;
          const bool triv = j1 == 0, raw = s == 1;
          if (!raw) {
            x[p0] = fe_scarry<P>(x[p0]);
            x[p0 + d2] = fe_scarry<P>(x[p0 + d2]);
            if (triv) { x[p0 + d1] = fe_scarry<P>(x[p0 + d1]); x[p0 + d2 + d1] = fe_scarry<P>(x[p0 + d2 + d1]); }
          }
          const int t1 = j1 << (lgn - s);
          sb(x[p0], x[p0 + d1], t1, triv);
          sb(x[p0 + d2], x[p0 + d2 + d1], t1, triv);
          if (triv && (!raw || lgn == 2)) x[p0 + d2] = fe_scarry<P>(x[p0 + d2]);
          sb(x[p0], x[p0 + d2], j1 << (lgn - s - 1), triv);
          sb(x[p0 + d1], x[p0 + d2 + d1], (j1 + half) << (lgn - s - 1), false);
        }
    }
    for (int k = 0; k < n; k++) {
      // what a strided pass stores (x * inter-pass twiddle, here the Montgomery one) must fit the packed words ...
      const Fe<P> st = fe_mul_sparse<P, true, 0>(fe_sbias<P>(x[k]), fe_one<P>());
      assert((i32)st.l[P::L - 1] >= 0 && st.l[P::L - 1] < (1u << (32 * P::NW - 29 * (P::L - 1))));
      // ... and what the last pass stores is canonical
      const Fe<P> c = fe_sreduce<P>(x[k]);
      assert(fe_eq_canon<P>(c, fe_reduce<P>(st)));
      fe_pack<P>(c, out + (size_t)k * P::NW);
    }
    return;
  }
  auto bfly = [&](Fe<P>& lo, Fe<P>& hi, int ti, bool trivial, bool raw, bool lazy) {
    Fe<P> t = hi;
    if (!trivial && use_shoup) t = fe_shoup_mul<P>(t, shoup + (size_t)ti * 18, shoup + (size_t)ti * 18 + 9);
    else if (!trivial) t = fe_mul<P>(t, tw[ti]);
    else if (!raw) t = fe_weak_reduce<P>(t);
    if (lazy) { hi = fe_sub<P, 8>(lo, t); lo = fe_add<P>(lo, t); }
    else { hi = fe_sub_carry<P, 8>(lo, t); lo = fe_add_carry<P>(lo, t); }
  };
  int s = 1;
  if (lgn & 1) {
    for (int g = 0; g < n / 2; g++) bfly(x[2 * g], x[2 * g + 1], 0, true, true, false);
    s = 2;
  }
  for (; s + 1 <= lgn; s += 2) {
    const int lgh = s - 1, half = 1 << lgh;
    use_shoup = shoup != nullptr && (lgn + lgc - 2 - lgh) >= 6;
    for (int grp = 0; grp < (n >> (s + 1)); grp++)
      for (int j1 = 0; j1 < half; j1++) {
        const int p0 = (grp << (s + 1)) | j1, d1 = half, d2 = half << 1;
        const bool triv = j1 == 0, raw = s == 1;
        const int t1 = j1 << (lgn - s);
        bfly(x[p0], x[p0 + d1], t1, triv, raw, true);
        bfly(x[p0 + d2], x[p0 + d2 + d1], t1, triv, raw, true);
        bfly(x[p0], x[p0 + d2], j1 << (lgn - s - 1), triv, false, false);
        bfly(x[p0 + d1], x[p0 + d2 + d1], (j1 + half) << (lgn - s - 1), false, false, false);
      }
  }
  for (int k = 0; k < n; k++) fe_pack<P>(fe_reduce<P>(x[k]), out + (size_t)k * P::NW);
}

extern "C" {
// fid: 0 = Fr, 1 = M128, 2 = Fq.  Words are the ABI encoding (8 or 4 u32, canonical).
int hc_field_op(int fid, int op, const u32* a, const u32* b, u32* out) {
  if (fid == 0) run_field<FrParams>(op, a, b, out);
  else if (fid == 1) run_field<M128Params>(op, a, b, out);
  else if (fid == 2) run_field<FqParams>(op, a, b, out);
  else return -1;
  return 0;
}
// one in-tile transform of 2^lgn points by the kernels' stage schedule (bounds asserted along the way)
int hc_ntt_tile(int fid, const u32* words, int lgn, const u32* tw_words, u32* out) {
  if (fid == 0) run_ntt_tile<FrParams>(words, lgn, tw_words, out);
  else if (fid == 1) run_ntt_tile<M128Params>(words, lgn, tw_words, out);
  else return -1;
  return 0;
}
// the same with the Shoup products of the wave-uniform stage pairs (BN254 Fr; lgc: log2 of the tile's columns in the kernel)
int hc_ntt_tile_shoup(const u32* words, int lgn, int lgc, const u32* tw_words, const u32* shoup, u32* out) {
  run_ntt_tile<FrParams>(words, lgn, tw_words, out, shoup, lgc);
  return 0;
}
// x * w mod p by fe_shoup_mul: x as 9 raw limbs (lazy forms allowed), w / wq as 9 limbs each; out = 9 limbs of the result
int hc_shoup_mul(const u32* x_limbs, const u32* w_limbs, const u32* wq_limbs, u32* out_limbs) {
  Fe<FrParams> x;
  for (int i = 0; i < 9; i++) x.l[i] = x_limbs[i];
  const Fe<FrParams> r = fe_shoup_mul<FrParams>(x, w_limbs, wq_limbs);
  for (int i = 0; i < 9; i++) out_limbs[i] = r.l[i];
  return 0;
}
// fe_mul_sparse<M128, signed, c> on raw limbs (a: 5 limbs, i32 when signed; b: 5 limbs below 2^29); out = 5 limbs
int hc_m128_mul_sparse(int is_signed, int c, const u32* a_limbs, const u32* b_limbs, u32* out_limbs) {
  Fe<M128Params> a, b, r;
  for (int i = 0; i < 5; i++) { a.l[i] = a_limbs[i]; b.l[i] = b_limbs[i]; }
  if (is_signed) r = c ? fe_mul_sparse<M128Params, true, 1>(a, b) : fe_mul_sparse<M128Params, true, 0>(a, b);
  else r = c ? fe_mul_sparse<M128Params, false, 1>(a, b) : fe_mul_sparse<M128Params, false, 0>(a, b);
  for (int i = 0; i < 5; i++) out_limbs[i] = r.l[i];
  return 0;
}
// fe_sreduce<M128> of 5 signed lazy limbs; out = 4 canonical words
int hc_m128_sreduce(const u32* a_limbs, u32* out_words) {
  Fe<M128Params> a;
  for (int i = 0; i < 5; i++) a.l[i] = a_limbs[i];
  fe_pack<M128Params>(fe_sreduce<M128Params>(a), out_words);
  return 0;
}
// pack(unpack(w)) round trip
int hc_pack_roundtrip(int fid, const u32* a, u32* out) {
  if (fid == 1) fe_pack<M128Params>(fe_unpack<M128Params>(a), out);
  else fe_pack<FrParams>(fe_unpack<FrParams>(a), out);
  return 0;
}
static Xyzz load_pt(const u32* w) {
  if (affine_words_is_inf(w)) return xyzz_inf();
  return xyzz_from_affine(affine_load_plain(w));
}
static void store_pt(const Xyzz& p, u32* w) {
  Affine a;
  if (!xyzz_to_affine(p, &a)) { memset(w, 0, 64); return; }
  affine_store_plain(a, w);
}
// op 0: madd (p XYZZ-ified + affine q), 1: add, 2: dbl(p), 3: dbl_affine(p), 4: signed madd p - q, 5: signed madd p + q
int hc_g1_op(int op, const u32* p, const u32* q, u32* out) {
  Xyzz P = load_pt(p), R;
  // scramble P's representation so ZZ != 1: P = (2P' - P') style is overkill; scale by lambda = 3:
  if (!xyzz_is_inf(P)) {
    Fq l = fe_to_mont<FqParams>(fe_unpack<FqParams>((const u32[]){3, 0, 0, 0, 0, 0, 0, 0}));
    Fq l2 = fe_sqr<FqParams>(l), l3 = fe_mul<FqParams>(l2, l);
    P.X = fe_mul<FqParams>(P.X, l2); P.Y = fe_mul<FqParams>(P.Y, l3);
    P.ZZ = fe_mul<FqParams>(P.ZZ, l2); P.ZZZ = fe_mul<FqParams>(P.ZZZ, l3);
  }
  switch (op) {
    case 0:
      if (affine_words_is_inf(q)) R = P; else R = xyzz_madd(P, affine_load_plain(q));
      break;
    case 1: R = xyzz_add(P, load_pt(q)); break;
    case 2: R = xyzz_dbl(P); break;
    case 3: R = affine_words_is_inf(p) ? xyzz_inf() : xyzz_dbl_affine(affine_load_plain(p)); break;
    case 4: case 5:
      if (affine_words_is_inf(q)) R = P; else R = xyzz_madd_signed_with<FeCpp>(P, affine_load_plain(q), op == 4);
      break;
    default: return -1;
  }
  store_pt(R, out);
  return 0;
}
// k * P by MSB-first double-and-madd over the kernel's XYZZ formulas (k: 8 u32 words)
int hc_g1_mul(const u32* p, const u32* k, u32* out) {
  Xyzz acc = xyzz_inf();
  if (!affine_words_is_inf(p)) {
    Affine a = affine_load_plain(p);
    for (int i = 255; i >= 0; i--) {
      acc = xyzz_dbl(acc);
      if ((k[i >> 5] >> (i & 31)) & 1) acc = xyzz_madd(acc, a);
    }
  }
  store_pt(acc, out);
  return 0;
}
// sum of n affine points through madd, storing/reloading the accumulator through the packed
// global-memory format every step (xyzz_store / xyzz_load)
int hc_g1_sum(const u32* pts, int n, u32* out) {
  u32 buf[32];
  xyzz_store(xyzz_inf(), buf);
  for (int i = 0; i < n; i++) {
    Xyzz acc = xyzz_load(buf);
    if (!affine_words_is_inf(pts + 16 * i)) acc = xyzz_madd(acc, affine_load_plain(pts + 16 * i));
    xyzz_store(acc, buf);
  }
  store_pt(xyzz_load(buf), out);
  return 0;
}
// ---- G2 (mzk_g2.h) with bounds assertions: op 0 add, 1 double, 2 scalar mul (k: 8 words), 3 Fq2 mul, 4 Fq2 inverse
int hc_g2_op(int op, const u32* a, const u32* b, const u32* k, u32* out) {
  if (op == 3 || op == 4) {
    Fq2 x = f2_load_plain(a), y = f2_load_plain(b);
    Fq2 r = (op == 3) ? f2_mul(x, y) : f2_inv(x);
    f2_store_plain(r, out);
    return 0;
  }
  Xyzz2 P = g2_words_is_inf(a) ? x2_inf() : x2_from_affine(g2_load_plain(a));
  Xyzz2 r;
  if (op == 0) {
    Xyzz2 Q = g2_words_is_inf(b) ? x2_inf() : x2_from_affine(g2_load_plain(b));
    // exercise both the mixed and the general addition
    Xyzz2 r1 = g2_words_is_inf(b) ? P : x2_madd(P, g2_load_plain(b));
    // general add on operands with non-trivial ZZ: (P + P) - P + Q style would change the value; instead re-randomise Z by doubling twice and halving is not available -> use x2_add directly
    r = x2_add(P, Q);
    u32 w1[32], w2[32];
    g2_store_plain(r1, w1); g2_store_plain(r, w2);
    for (int i = 0; i < 32; i++) if (w1[i] != w2[i]) return -7;
  } else if (op == 1) {
    r = x2_dbl(P);
  } else if (op == 2) {
    if (g2_words_is_inf(a)) r = x2_inf(); else r = x2_scalar_mul(g2_load_plain(a), k);
  } else return -1;
  // round trip through the XYZZ record as the kernels do
  u32 rec[64];
  x2_store(r, rec);
  g2_store_plain(x2_load(rec), out);
  return 0;
}
// GLV split (mzk_glv.h): out = m1 (4 words) | neg1 | m2 (4 words) | neg2
int hc_glv_split(const u32* k, u32* out) {
  glv_split(k, out, out + 4, out + 5, out + 9);
  return 0;
}
}
