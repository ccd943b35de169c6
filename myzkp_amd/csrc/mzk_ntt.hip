// mzk_ntt.hip -- radix-2 NTT / iNTT / coset LDE on gfx950.
//
// Stands behind ntt::ntt / ntt::intt (myzkp/src/modules/algebra/ntt.rs:7-64), the transform inside
// Polynomial::fft_multiply (polynomial.rs:242-300) and ntt::fast_multiply (ntt.rs:66-116), and
// ntt::fast_coset_evaluate (ntt.rs:254-269).  Same map as the reference: natural order in, natural
// order out, out[k] = sum_j in[j] root^(j k).
//
// Algorithm: n = n_1 n_2 ... n_K (K <= 4, n_t <= 2^8, chosen so that every
// workgroup tile is 1024 elements).  Input index j = (j_1, ..., j_K) row-major, output index
// k = k_1 + n_1 (k_2 + n_2 (...)).  Pass t < K transforms along j_t in place on strided columns
// (tile = n_t x C contiguous columns, C * 32 B runs), and multiplies element (k_t, j') of each block by
// the inter-pass twiddle w_t^(j' k_t) read COALESCED from a table laid out like the data.  The last
// pass transforms contiguous rows of n_K and scatters them transposed so that R = 1024 / n_K
// consecutive outputs are written together.  No bit-reversal pass over HBM: the permutation happens
// inside LDS.
//
// Arithmetic trick: values stay in the plain (non-Montgomery) domain.  Every multiplication in an
// NTT is by a constant (twiddle, n^-1, offset^i); with the constant in Montgomery form w R,
// fe_mul(x, wR) = x w exactly, so no element is ever converted into or out of Montgomery form.
//
// LDS layout: structure-of-arrays, limb-major (lds[limb][position]) so every access is a
// ds_read_b32 / ds_write_b32; positions are XOR-swizzled (Geo::phys) so that neither the bit-reversed scatter of the
// load phase nor the strided stage accesses serialise on a bank.
#include <stdlib.h>
#include "mzk_common.h"
#include "mzk_field_asm.h"

namespace mzk {

constexpr int TILE_LOG = 10;
#ifndef MZK_NTT_THREADS
#define MZK_NTT_THREADS 256
#endif
constexpr int NTHREADS = MZK_NTT_THREADS;
#ifndef MZK_NTT_MAX_LEVEL_LOG
#define MZK_NTT_MAX_LEVEL_LOG 8
#endif
constexpr int MAX_LEVEL_LOG = MZK_NTT_MAX_LEVEL_LOG;
#ifndef MZK_NTT_LAZY_FIRST
#define MZK_NTT_LAZY_FIRST 1       // 0: A/B builds of the carry-everywhere butterflies (tools/timing/time_ntt.py with MZK_HIP_LIB)
#endif
constexpr bool NTT_LAZY_FIRST = MZK_NTT_LAZY_FIRST != 0;
#ifndef MZK_NTT_EARLY_TW
#define MZK_NTT_EARLY_TW 0         // 1: A/B builds with the inter-pass twiddles requested at the top of the strided pass (measured slower, see k_ntt_strided)
#endif
// Tile geometry.  Small (1024 elements, 256 lanes, levels of <= 2^8): every size below 2^20.  Large (4096 elements,
// 1024 lanes = one workgroup per CU, levels of <= 2^10): from 2^20 points on, where 256+ workgroups exist -- a 2^20
// transform is TWO passes of 2^10 levels instead of three (one global round trip and one inter-pass twiddle product
// per element less), each tile still 4+ adjacent columns wide (>= 128-byte runs).  The Fr tile is 144 KiB of limbs, so
// its in-tile twiddles are staged as packed words (16 KiB: 160 KiB exactly) and unpacked at use.
// TWG_ (M128 large tiles): the in-tile twiddles are NOT staged in LDS -- a 4096-element M128 tile is 80 KiB, so without its
// 10 KiB of twiddles TWO workgroups of 512 lanes share a CU and one's loads / stores run under the other's butterflies (the
// single 1024-lane workgroup per CU had nothing to overlap its ~22 us of global traffic with).  A twiddle is 16 packed bytes:
// stage pairs in which a wave shares its twiddles read them by scalar loads, the others by one dwordx4 per twiddle from L1.
// WPE_: waves per SIMD the register budget must allow (4 = two 512-lane workgroups per CU).
template <int TL_, int NT_, int MAXLV_, bool TWG_ = false, int WPE_ = 1> struct Geo {
  static constexpr int TL = TL_, TILE = 1 << TL_, NT = NT_, MAXLV = MAXLV_, WPE = WPE_;
  static constexpr bool TWG = TWG_;
  static constexpr int GQ = TILE / NT_ / 4;            // radix-4 groups per lane and stage pair (fused first / last pairs)
  // in-tile twiddles in LDS: entries per limb / word row.  A compile-time stride (the largest level's n/2, whatever this pass's level
  // is) puts the rows at immediate offsets of ONE address: four vector additions per twiddle fetch less than a runtime stride.
  static constexpr int TWS = 1 << (MAXLV_ - 1);
  // Logical tile position -> LDS word index inside a limb row.  The low five bits (the bank) are XORed with a GF(2)-linear
  // function of the upper bits, chosen by simulating every wave-level access of the kernels (bit-reversed scatter of
  // the load phase, the four loads and stores of every radix-4 stage pair, all level sizes the geometry runs;
  // tools/timing/lds_swizzle_search.py) so that each 32-lane half of an access touches every bank at most once:
  // small tiles, 2^7 and 2^8 levels: 288 cycles per tile (conflict-free) instead of 544 with the round-1 swizzle
  // pos ^ (pos >> 5 & 31); large tiles, 2^10 levels: 1408 (conflict-free) instead of 3840.  (Measured effect on the
  // transform time: none -- the LDS traffic hides behind the products; what-if builds in DESIGN.md section 4.)
  static __device__ __forceinline__ int phys(int pos) {
    const int h = pos >> 5;
    const int x = (TL_ == 10) ? (h ^ (h << 2) ^ (h << 3)) : ((h >> 2) ^ (h << 1) ^ (h << 3));
    return pos ^ (x & 31);
  }
  template <class P> static constexpr bool twpack() { return !TWG_ && (size_t)4 * P::L * (TILE + (1 << (MAXLV_ - 1))) > (size_t)160 * 1024; }
  template <class P> static constexpr size_t lds_bytes(int lgn) {
    if (TWG_) return sizeof(u32) * (size_t)P::L * TILE;
    return sizeof(u32) * ((size_t)P::L * TILE + (size_t)(twpack<P>() ? P::NW : P::L) * TWS);
  }
};
typedef Geo<TILE_LOG, NTHREADS, MAX_LEVEL_LOG> GeoS;
typedef Geo<TILE_LOG, NTHREADS, TILE_LOG> GeoS1;   // the one-pass transforms of 2^9 and 2^10 points: their level is wider than GeoS's twiddle rows
typedef Geo<12, 1024, 10> GeoL;
typedef Geo<12, 512, 10, true, 4> GeoM;        // M128 large tiles: two workgroups per CU
template <class P> struct LargeGeo { typedef GeoL type; };
template <> struct LargeGeo<M128Params> { typedef GeoM type; };
// The large geometry is used exactly where it saves a whole pass: 2^20 (two passes of 2^10 instead of 7/7/6: measured
// Fr 0.147 -> 0.137 ms, M128 0.063 -> 0.056 ms) and 2^25 .. 2^30.  Where both geometries need three passes the small
// tiles win (2^21 .. 2^24: Fr +8 %, M128 +16 % with large tiles; tools/timing/time_ntt.py with MZK_NTT_LARGE_FR /
// MZK_NTT_LARGE_M128 = smallest log2 size forced onto the large geometry, 99 = never).
// A BATCH of transforms gives the small tiles what a single 2^20 transform lacks: several rounds of workgroups per CU,
// so the loads and stores of one tile run under the butterflies of its neighbours (four small-tile workgroups share a
// CU; the large tile owns it) -- from three Fr / two M128 transforms of 2^20 on, three overlapped passes beat two
// exposed ones (tools/timing/ntt_batch.py: Fr 0.1045 vs 0.1131 ms per transform at batch 16, M128 0.041 vs 0.048).
static bool large_geo(int fid, unsigned logn, size_t batch = 1) {
  static const int env_fr = tune_int("MZK_NTT_LARGE_FR", -1);
  static const int env_m = tune_int("MZK_NTT_LARGE_M128", -1);
  const int env = fid == MZK_FIELD_M128 ? env_m : env_fr;
  if (env >= 0) return logn >= (unsigned)env && logn >= 14;
  if (logn < 20) return false;
  if (batch > (fid == MZK_FIELD_M128 ? 1u : 2u)) return false;
  return (logn + 9) / 10 < (logn + MAX_LEVEL_LOG - 1) / MAX_LEVEL_LOG;
}

struct Words8 { u32 w[8]; };

struct LevelInfo {
  int nlev;
  int lg[4];
};

// ---- global memory <-> limbs ----------------------------------------------------------------------
// The streaming accesses of the transform passes -- the data (read once, written once per pass) and the inter-pass twiddles -- can
// carry the non-temporal hint (MZK_NTT_NT = 1).  A plain copy gains 25 % from it on MI355X (profiles/round5_copy_kernel_variants.txt);
// the transforms LOSE: M128 2^20 0.0407 -> 0.0433 ms, 2^22 0.151 -> 0.180 (same box, profiles/round5_ntt_nontemporal_ab.txt) -- what
// one pass writes, the next one reads back out of L2 / the 256-MiB Infinity Cache, and so it does the twiddle tables of the
// previous transform; the hint takes that away.  Off.
#ifndef MZK_NTT_NT
#define MZK_NTT_NT 0
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t stream_load16(const u32* p) {
  if (MZK_NTT_NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return *reinterpret_cast<const u32x4_t*>(p);
}
// wt: the store goes out write-through at agent scope (`sc1`), so what a pass writes reaches the fabric while the other waves
// still compute instead of sitting dirty in the XCDs' L2s until the end-of-kernel release flushes it.  A template flag of the pass
// kernels (a run-time branch around the store cost the 128-VGPR instantiations two to four spilled registers), chosen per transform
// size by run_plan_geo.
__device__ __forceinline__ void stream_store16(u32* p, u32x4_t v, bool wt) {
  if (MZK_NTT_NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
  else if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else *reinterpret_cast<u32x4_t*>(p) = v;
}
template <class P> __device__ __forceinline__ void gload_words_stream(const u32* __restrict__ g, size_t idx, u32 (&w)[P::NW]) {
#pragma unroll
  for (int q = 0; q < P::NW / 4; q++) {
    const u32x4_t v = stream_load16(g + idx * P::NW + 4 * q);
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
}
template <class P> __device__ __forceinline__ void gstore_stream(u32* __restrict__ g, size_t idx, const Fe<P>& v, bool wt) {
  u32 w[P::NW];
  fe_pack<P>(v, w);
#pragma unroll
  for (int q = 0; q < P::NW / 4; q++) {
    u32x4_t t;
    t.x = w[4 * q]; t.y = w[4 * q + 1]; t.z = w[4 * q + 2]; t.w = w[4 * q + 3];
    stream_store16(g + idx * P::NW + 4 * q, t, wt);
  }
}
template <class P> __device__ __forceinline__ Fe<P> gload(const u32* __restrict__ g, size_t idx) {
  u32 w[P::NW];
  const uint4* p4 = reinterpret_cast<const uint4*>(g + idx * P::NW);
#pragma unroll
  for (int q = 0; q < P::NW / 4; q++) {
    uint4 v = p4[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  return fe_unpack<P>(w);
}
// the packed words only: the tile loops issue the loads of ALL their elements first and unpack afterwards, so a lane has
// several 32-byte requests in flight instead of one per loop trip (load, wait, store to LDS, next)
template <class P> __device__ __forceinline__ void gload_words(const u32* __restrict__ g, size_t idx, u32 (&w)[P::NW]) {
  const uint4* p4 = reinterpret_cast<const uint4*>(g + idx * P::NW);
#pragma unroll
  for (int q = 0; q < P::NW / 4; q++) {
    uint4 v = p4[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
}
template <class P> __device__ __forceinline__ void gstore(u32* __restrict__ g, size_t idx, const Fe<P>& v) {
  u32 w[P::NW];
  fe_pack<P>(v, w);
  uint4* p4 = reinterpret_cast<uint4*>(g + idx * P::NW);
#pragma unroll
  for (int q = 0; q < P::NW / 4; q++) p4[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
// value < 2p, normalised -> something fe_pack can hold (value < 2^(32 NW)).
template <class P> __device__ __forceinline__ Fe<P> fe_fit(const Fe<P>& v) {
  if constexpr (P::BITS + 1 > 32 * P::NW) return fe_cond_sub_p<P>(v);
  else return v;
}

// What a pass stores: x * (inter-pass twiddle) as something fe_pack can hold.  Signed lazy x: made non-negative first (three
// additions), so that the signed product returns a value in [0, 1.05 p + 3) -- below 2^128 as it stands.
template <class P> __device__ __forceinline__ Fe<P> inter_mul(const Fe<P>& x, const Fe<P>& tw) {
  if constexpr (SparseMod<P>::value) return FeAsm<P>::smul(fe_sbias<P>(x), tw);
  else return fe_fit<P>(FeAsm<P>::mul(x, tw));
}
// the last pass's output: optional scale, canonical representative
template <class P> __device__ __forceinline__ Fe<P> final_reduce(const Fe<P>& x, const Fe<P>& sc, int has_scale) {
  if constexpr (SparseMod<P>::value) {
    Fe<P> v = x;
    if (has_scale) v = FeAsm<P>::smul(v, sc);
    return fe_sreduce<P>(v);
  } else {
    Fe<P> v = x;
    if (has_scale) v = FeAsm<P>::mul(v, sc);
    return fe_reduce<P>(v);
  }
}

template <class P, class G> __device__ __forceinline__ Fe<P> lds_load(const u32* lds, int pos) {
  Fe<P> r;
  const int ph = G::phys(pos);
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = lds[i * G::TILE + ph];
  return r;
}
template <class P, class G> __device__ __forceinline__ void lds_store(u32* lds, int pos, const Fe<P>& v) {
  const int ph = G::phys(pos);
#pragma unroll
  for (int i = 0; i < P::L; i++) lds[i * G::TILE + ph] = v.l[i];
}
// The same by PHYSICAL index.  Geo::phys is GF(2)-linear (shifts and XORs of the position's bits), so the four positions of a
// radix-4 group, p0 + d with d in {0, d1, d2, d1 + d2} and p0 zero at the bits of d1 and d2, are phys(p0) ^ phys(d): one swizzle per
// group and three XORs with wave-uniform constants instead of four swizzles (7 instructions each).
template <class P, class G> __device__ __forceinline__ Fe<P> lds_load_ph(const u32* lds, int ph) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = lds[i * G::TILE + ph];
  return r;
}
template <class P, class G> __device__ __forceinline__ void lds_store_ph(u32* lds, int ph, const Fe<P>& v) {
#pragma unroll
  for (int i = 0; i < P::L; i++) lds[i * G::TILE + ph] = v.l[i];
}

// In-tile twiddles w_n^j (j < n/2) staged in LDS in limb form, limb-major, so a butterfly fetches its
// twiddle with L ds_read_b32 and no unpacking.
template <class P, class G>
__device__ __forceinline__ void stage_twiddles(u32* twl, const u32* __restrict__ tw, int lgn) {
  if constexpr (G::TWG) return;            // read from global memory at use (bfly)
  const int cnt = (lgn >= 2) ? (1 << (lgn - 1)) : 0;
  for (int j = threadIdx.x; j < cnt; j += G::NT) {
    if constexpr (G::template twpack<P>()) {      // packed words, word-major
#pragma unroll
      for (int i = 0; i < P::NW; i++) twl[i * G::TWS + j] = tw[(size_t)j * P::NW + i];
    } else {
      Fe<P> w = gload<P>(tw, j);
#pragma unroll
      for (int i = 0; i < P::L; i++) twl[i * G::TWS + j] = w.l[i];
    }
  }
}
// One DIT butterfly in registers: (lo, hi) <- (lo + w hi, lo - w hi).  `trivial` (w = 1) skips the product;
// then hi must still be brought below the 4 p the K = 8 subtraction tolerates, unless it is a raw input
// (stage 1: < 2^(32 NW)).
// LAZY: the sums stay limb-wise (no carry propagation) -- the first stage of a register-resident stage pair, whose outputs
// are only added, subtracted or multiplied again before the second stage normalises them (limbs < 2^29 + 2^30 there: the
// product takes one operand with limbs up to 3 * 2^30, fe_mul's column bound, and the carrying add/sub any u32 that
// does not overflow with 8p added).  Same values either way.
// SIGNED LAZY butterflies (sparse-modulus fields: M128).  Values are sums of i32 limbs (mzk_field.h, fe_sadd / fe_ssub): a butterfly
// is the product plus ONE instruction per limb and output -- no K p, no carry chain (10 instructions where the carrying pair took
// 31 and the limb-wise one 15).  What keeps the limbs inside i32: the signed product returns limbs in [0, 2^29), and an operand that
// is NOT multiplied on entry to a stage pair is carried once (fe_scarry: 12 instructions) -- x0 and x2 of a radix-4 group, all four
// when the group's twiddles are 1.  With B = 2^29 - 1: inputs of a pair <= B  =>  after the first stage |.| <= 2 B, after the second
// <= 3 B (4 B for the all-ones group of a raw first pair: x0 + x1 + x2 + x3), stored to LDS as they are; values only double per level (no reduction
// mod p on the way), so after the <= 10 levels of a pass |x| < 2^138, which the products and the final reductions take
// (fe_sbias / fe_sreduce: < 2^12 p).  tests/hostcheck restates the schedule with every bound as an assertion.
template <class P> struct SignedLazy { static constexpr bool value = SparseMod<P>::value; };
template <class P, class G>
__device__ __forceinline__ Fe<P> tw_fetch(const u32* twl, int tws, int ti) {
  Fe<P> w;
  if constexpr (G::TWG) {
    w = gload<P>(twl, (size_t)ti);         // twl = the plan's table in global memory (16 bytes per M128 twiddle; L1 / scalar cache)
  } else if constexpr (G::template twpack<P>()) {
    u32 ww[P::NW];
#pragma unroll
    for (int i = 0; i < P::NW; i++) ww[i] = twl[i * G::TWS + ti];
    w = fe_unpack<P>(ww);
  } else {
#pragma unroll
    for (int i = 0; i < P::L; i++) w.l[i] = twl[i * G::TWS + ti];
  }
  return w;
}
template <class P, class G>
__device__ __forceinline__ void sbfly(Fe<P>& lo, Fe<P>& hi, const u32* twl, int tws, int ti, bool trivial) {
  Fe<P> t = hi;
  if (!trivial) t = FeAsm<P>::smul(t, tw_fetch<P, G>(twl, tws, ti));
  hi = fe_ssub<P>(lo, t);
  lo = fe_sadd<P>(lo, t);
}
template <class P, class G, bool LAZY = false>
__device__ __forceinline__ void bfly(Fe<P>& lo, Fe<P>& hi, const u32* twl, int tws, int ti, bool trivial, bool raw) {
  if constexpr (SignedLazy<P>::value) { sbfly<P, G>(lo, hi, twl, tws, ti, trivial); return; }
  Fe<P> t = hi;
  if (!trivial) {
    t = FeAsm<P>::mul(t, tw_fetch<P, G>(twl, tws, ti));
  } else if (!raw) {
    t = fe_weak_reduce<P>(t);
  }
  if constexpr (LAZY) {
    hi = fe_sub<P, 8>(lo, t);
    lo = fe_add<P>(lo, t);
  } else {
    hi = fe_sub_carry<P, 8>(lo, t);
    lo = fe_add_carry<P>(lo, t);
  }
}
// In-LDS radix-2/radix-4 DIT over the k-dimension of a [2^lgn][2^lgc] tile whose rows were stored
// bit-reversed; leaves natural order.  Stages run in pairs: a lane loads the four elements of a radix-4
// group once, does two butterfly stages in registers and stores them (half the LDS traffic, address
// arithmetic and barriers of one-stage-at-a-time); an odd lgn starts with one radix-2 stage.  Groups are
// enumerated twiddle-major (all groups with twiddle index 0 first), so whole waves skip the products by
// w^0 = 1: about one stage's worth of products per level.
// One radix-4 group in registers: the two butterfly stages s, s + 1 on the four elements at k-distance 2^(s-1), 2^s
// (x0, x1 = first stage pair; x2, x3 the second); j1 = the group's index inside the first stage's half.
// MODE: 0 = a lane decides for itself whether its twiddles are 1 (execution-masked branches around every product and register
// copies where they join); 1 = no lane of the wave has j1 = 0, 2 = all have: straight-line code.  radix4_wave picks the mode from a
// ballot -- all but one or two waves of a stage pair are uniform (the groups are enumerated twiddle-major).
template <class P, class G, int MODE = 0>
__device__ __forceinline__ void radix4_regs(Fe<P>& x0, Fe<P>& x1, Fe<P>& x2, Fe<P>& x3, const u32* twl, int tws, int lgn, int s, int j1) {
  const int lgh = s - 1;
  const bool triv = MODE == 0 ? (j1 == 0) : (MODE == 2);
  const bool raw = (s == 1);
  const int t1 = j1 << (lgn - s);
  if constexpr (SignedLazy<P>::value) {
    if (!raw) {          // raw inputs (stage 1) are unpacked words: carried already
      x0 = fe_scarry<P>(x0);
      x2 = fe_scarry<P>(x2);
      if (triv) { x1 = fe_scarry<P>(x1); x3 = fe_scarry<P>(x3); }
    }
  }
  bfly<P, G, NTT_LAZY_FIRST>(x0, x1, twl, tws, t1, triv, raw);
  bfly<P, G, NTT_LAZY_FIRST>(x2, x3, twl, tws, t1, triv, raw);
  if constexpr (SignedLazy<P>::value) {
    // the all-ones group would end at x0 + x1 + x2 + x3 = 4 B; only raw first pairs may (their readers carry or multiply first):
    // everywhere else x2 + x3 is carried here, so that what reaches the final reductions stays at 3 B (1 group in 2^(s-1))
    if (triv && (!raw || lgn == 2)) x2 = fe_scarry<P>(x2);
  }
  bfly<P, G>(x0, x2, twl, tws, j1 << (lgn - s - 1), triv, false);
  bfly<P, G>(x1, x3, twl, tws, (j1 + (1 << lgh)) << (lgn - s - 1), false, false);
}
template <class P, class G>
__device__ __forceinline__ void radix4_wave(Fe<P>& x0, Fe<P>& x1, Fe<P>& x2, Fe<P>& x3, const u32* twl, int tws, int lgn, int s, int j1) {
  // (M128 single-workgroup tiles only: three copies of the group are ~30 more VGPRs, which the BN254 kernels and the 512-lane M128 form -- both at their 128-register cap -- would spill)
  if constexpr (!SignedLazy<P>::value || G::TWG) { radix4_regs<P, G, 0>(x0, x1, x2, x3, twl, tws, lgn, s, j1); return; }
  const unsigned long long trivial = __ballot(j1 == 0);
  if (trivial == 0) radix4_regs<P, G, 1>(x0, x1, x2, x3, twl, tws, lgn, s, j1);
  else if (trivial == __ballot(1)) radix4_regs<P, G, 2>(x0, x1, x2, x3, twl, tws, lgn, s, j1);
  else radix4_regs<P, G, 0>(x0, x1, x2, x3, twl, tws, lgn, s, j1);
}
// Stage pairs whose twiddles are the same for a whole wave (groups are enumerated twiddle-major: 2^lgrest consecutive groups
// share j1, so from lgrest >= 6 on a wave has ONE j1) take them from the plan's Shoup table by scalar loads -- entry ti =
// 32 words: limbs of the plain w^ti at [0, 9), of floor(w^ti 2^261 / p) at [16, 25) -- and multiply by the precomputed-quotient
// product (fe_shoup_mul: 143 multiply-adds, the constant in scalar registers) instead of the Montgomery one (162 + the twiddle's
// eight LDS reads and its unpacking).  BN254 Fr only (M128's sparse modulus makes its Montgomery reduction cheaper than that).
constexpr int SHOUP_ENTRY_WORDS = 32;
template <class P> struct HasShoup { static constexpr bool value = false; };
template <> struct HasShoup<FrParams> { static constexpr bool value = true; };
template <class P, bool LAZY>
__device__ __forceinline__ void bfly_shoup(Fe<P>& lo, Fe<P>& hi, const u32* __restrict__ entry, bool trivial, bool raw) {
  Fe<P> t = hi;
  if (!trivial) {
    u32 w[P::L], wq[P::L];
#pragma unroll
    for (int i = 0; i < P::L; i++) { w[i] = entry[i]; wq[i] = entry[16 + i]; }
    t = FeAsm<P>::shoup_mul(t, w, wq);
  } else if (!raw) {
    t = fe_weak_reduce<P>(t);
  }
  if constexpr (LAZY) {
    hi = fe_sub<P, 8>(lo, t);
    lo = fe_add<P>(lo, t);
  } else {
    hi = fe_sub_carry<P, 8>(lo, t);
    lo = fe_add_carry<P>(lo, t);
  }
}
// radix4_regs with a wave-uniform j1 (an SGPR: the table addresses are scalar)
template <class P>
__device__ __forceinline__ void radix4_shoup(Fe<P>& x0, Fe<P>& x1, Fe<P>& x2, Fe<P>& x3, const u32* __restrict__ tab, int lgn, int s, int j1) {
  const int lgh = s - 1;
  const bool triv = (j1 == 0);
  const bool raw = (s == 1);
  const u32* e1 = tab + (size_t)(j1 << (lgn - s)) * SHOUP_ENTRY_WORDS;
  const u32* e2 = tab + (size_t)(j1 << (lgn - s - 1)) * SHOUP_ENTRY_WORDS;
  const u32* e3 = tab + (size_t)((j1 + (1 << lgh)) << (lgn - s - 1)) * SHOUP_ENTRY_WORDS;
  bfly_shoup<P, NTT_LAZY_FIRST>(x0, x1, e1, triv, raw);
  bfly_shoup<P, NTT_LAZY_FIRST>(x2, x3, e1, triv, raw);
  bfly_shoup<P, false>(x0, x2, e2, triv, false);
  bfly_shoup<P, false>(x1, x3, e3, false, false);
}

// The first stage pair (s = 1) of a tile with an even number of levels on elements that never went through LDS: a lane's four
// loads ARE a radix-4 group (rows r + u 2^(lgn-2) land on bit-reversed rows 4 brev(r) + brev2(u)).  All twiddles are 1 except
// w^(n/4), taken from the global table (the LDS copy is not staged yet: no barrier in front of this).
template <class P, class G>
__device__ __forceinline__ void first_pair_regs(Fe<P>& x0, Fe<P>& x1, Fe<P>& x2, Fe<P>& x3, const u32* __restrict__ tw_tile, const u32* __restrict__ tw_shoup,
                                                int lgn) {
  if (lgn & 1) {        // odd level count: the lone radix-2 stage (all twiddles 1) -- the lane holds two of its butterflies
    bfly<P, G>(x0, x1, nullptr, 0, 0, true, true);
    bfly<P, G>(x2, x3, nullptr, 0, 0, true, true);
    return;
  }
  if constexpr (HasShoup<P>::value) {
    if (tw_shoup) { radix4_shoup<P>(x0, x1, x2, x3, tw_shoup, lgn, 1, 0); return; }
  }
  const Fe<P> zeta = gload<P>(tw_tile, (size_t)1 << (lgn - 2));
  bfly<P, G, NTT_LAZY_FIRST>(x0, x1, nullptr, 0, 0, true, true);
  bfly<P, G, NTT_LAZY_FIRST>(x2, x3, nullptr, 0, 0, true, true);
  bfly<P, G>(x0, x2, nullptr, 0, 0, true, false);
  if constexpr (SignedLazy<P>::value) {
    const Fe<P> t = FeAsm<P>::smul(x3, zeta);
    x3 = fe_ssub<P>(x1, t);
    x1 = fe_sadd<P>(x1, t);
  } else {
    const Fe<P> t = FeAsm<P>::mul(x3, zeta);
    x3 = fe_sub_carry<P, 8>(x1, t);
    x1 = fe_add_carry<P>(x1, t);
  }
}
// fused first / last stage pairs need one radix-4 group per lane (a full tile) and at least one stage pair after the first stage(s)
template <class G> __device__ __forceinline__ bool fuse_edges(int lgn, int lgc, int enable) { return enable && lgn >= 4 && (lgn + lgc) == G::TL; }

// In-LDS radix-2/radix-4 DIT over the k-dimension of a [2^lgn][2^lgc] tile whose rows were stored
// bit-reversed; leaves natural order.  Stages run in pairs: a lane loads the four elements of a radix-4
// group once, does two butterfly stages in registers and stores them (half the LDS traffic, address
// arithmetic and barriers of one-stage-at-a-time); an odd lgn starts with one radix-2 stage.  Groups are
// enumerated twiddle-major (all groups with twiddle index 0 first), so whole waves skip the products by
// w^0 = 1: about one stage's worth of products per level.
// Stages s_from .. s_to only (pairs; s_from = 3 / s_to = lgn - 2 when the kernel runs the first / last pair itself on
// registers next to its global loads / stores).
template <class P, class G>
__device__ __forceinline__ void tile_stages(u32* lds, const u32* twl, int lgn, int lgc, int s_from, int s_to, const u32* __restrict__ tw_shoup = nullptr) {
  const int tid = threadIdx.x;
  const int cmask = (1 << lgc) - 1;
  const int tws = (lgn >= 2) ? (1 << (lgn - 1)) : 1;   // twiddle table stride (entries per limb row)
  int s = s_from;
  if (s == 1 && (lgn & 1)) {  // stage 1 alone: every twiddle is 1
    const int nbf = 1 << (lgn + lgc - 1);
    for (int b = tid; b < nbf; b += G::NT) {
      const int c = b & cmask, grp = b >> lgc;
      const int plo = ((grp << 1) << lgc) | c, phi = plo + (1 << lgc);
      Fe<P> x0 = lds_load<P, G>(lds, plo), x1 = lds_load<P, G>(lds, phi);
      bfly<P, G>(x0, x1, twl, tws, 0, true, true);
      lds_store<P, G>(lds, plo, x0);
      lds_store<P, G>(lds, phi, x1);
    }
    __syncthreads();
    s = 2;
  }
  const int lgg = lgn + lgc - 2;   // log2(radix-4 groups per stage pair)
  for (; s + 1 <= s_to; s += 2) {
    const int lgh = s - 1;                 // log2 of the first stage's half-distance (in k)
    const int lgrest = lgg - lgh;
    for (int g = tid; g < (1 << lgg); g += G::NT) {
      const int j1 = g >> lgrest;
      const int rest = g & ((1 << lgrest) - 1);
      const int c = rest & cmask, grp = rest >> lgc;
      const int k0 = (grp << (s + 1)) | j1;
      const int p0 = (k0 << lgc) | c;
      const int d1 = 1 << (lgh + lgc), d2 = d1 << 1;
      const int ph0 = G::phys(p0), ph1 = ph0 ^ G::phys(d1), ph2 = ph0 ^ G::phys(d2), ph3 = ph1 ^ G::phys(d2);
      Fe<P> x0 = lds_load_ph<P, G>(lds, ph0), x1 = lds_load_ph<P, G>(lds, ph1);
      Fe<P> x2 = lds_load_ph<P, G>(lds, ph2), x3 = lds_load_ph<P, G>(lds, ph3);
      bool done = false;
      if constexpr (HasShoup<P>::value) {
        if (tw_shoup && lgrest >= 6) {       // one j1 per wave: scalar twiddles
          radix4_shoup<P>(x0, x1, x2, x3, tw_shoup, lgn, s, __builtin_amdgcn_readfirstlane(j1));
          done = true;
        }
      }
      if (!done) {
        if (G::TWG && lgrest >= 6) radix4_wave<P, G>(x0, x1, x2, x3, twl, tws, lgn, s, __builtin_amdgcn_readfirstlane(j1));   // scalar loads
        else radix4_wave<P, G>(x0, x1, x2, x3, twl, tws, lgn, s, j1);
      }
      lds_store_ph<P, G>(lds, ph0, x0);
      lds_store_ph<P, G>(lds, ph1, x1);
      lds_store_ph<P, G>(lds, ph2, x2);
      lds_store_ph<P, G>(lds, ph3, x3);
    }
    __syncthreads();
  }
}
template <class P, class G>
__device__ __forceinline__ void tile_stages(u32* lds, const u32* twl, int lgn, int lgc) { tile_stages<P, G>(lds, twl, lgn, lgc, 1, lgn); }

// Pass t < K: in-place strided columns + inter-pass twiddle.
// PRE (first pass of a coset LDE, ntt.rs:254-269): the input is the coefficient vector itself -- element idx is
// coef[idx] * offset^idx for idx < n_coef and zero beyond -- so Polynomial::scale and the zero padding cost no pass
// of their own.  offset^idx = offset^(j1 M) * offset^col comes from two small per-call tables (pre_row: 2^lgn
// entries, pre_col: M entries, Montgomery form).
struct PreArgs { const u32* coef; size_t n_coef; const u32* pre_row; const u32* pre_col; };
template <class P, bool PRE, class G, bool WT>
__global__ __launch_bounds__(G::NT, G::WPE) void k_ntt_strided(const u32* __restrict__ in, u32* __restrict__ out,
                                                           const u32* __restrict__ tw_tile,
                                                           const u32* __restrict__ tw_inter, int lgn, int lgM, int lgc, PreArgs pre, int fuse, const u32* __restrict__ tw_shoup) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const int tid = threadIdx.x;
  const int lg_tiles = lgM - lgc;
  const size_t o = (size_t)blockIdx.x >> lg_tiles;
  const size_t ct = (size_t)blockIdx.x & (((size_t)1 << lg_tiles) - 1);
  const size_t base = (o << (lgn + lgM)) + (ct << lgc);
  const int cmask = (1 << lgc) - 1;
  const int tile_elems = 1 << (lgn + lgc);
  const u32* twl = G::TWG ? tw_tile : lds + P::L * G::TILE;
  stage_twiddles<P, G>(lds + P::L * G::TILE, tw_tile, lgn);
  constexpr int UNR = G::TILE / G::NT;          // elements per lane and tile (tile_elems == G::TILE here)
  constexpr int GQ = G::GQ;                     // radix-4 groups per lane: element u = q + GQ v is member v of group q
  static_assert(UNR == 4 * GQ, "whole radix-4 groups per lane");
  constexpr bool wt = WT;
  const bool fused = fuse_edges<G>(lgn, lgc, fuse);   // first (plain loads only) and last stage pair on registers, next to the global accesses
  // EARLY_TW (M128, one 1024-lane workgroup per CU: 44 of 128 VGPRs in use): the inter-pass twiddles of the lane's four elements are
  // requested right behind its data, at the top of the kernel, and wait in 16 registers.  A 2^20 transform is one tile per CU with
  // all CUs in lock-step, so the request after the stage loop met an idle HBM and every wave waited out the whole 16-MiB burst
  // (~4 us of a 24-us pass); now that burst streams in under the butterflies.
  constexpr bool EARLY_OK = SparseMod<P>::value && G::NT == 1024 && !G::TWG;
  constexpr bool EARLY_TW = MZK_NTT_EARLY_TW == 1 && EARLY_OK;          // at the top, behind the data loads
  constexpr bool MID_TW = MZK_NTT_EARLY_TW == 2 && EARLY_OK;            // behind the first barrier: the data has arrived, HBM is idle
  u32 tw[UNR][P::NW];
  auto tw_load_at = [&](int t, int u) {
    const int e = t + u * G::NT;
    gload_words_stream<P>(tw_inter, ((size_t)(e >> lgc) << lgM) + (ct << lgc) + (e & cmask), tw[u]);
  };
  if constexpr (!PRE) {
    u32 w[UNR][P::NW];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int e = tid + u * G::NT;
      gload_words_stream<P>(in, base + ((size_t)(e >> lgc) << lgM) + (e & cmask), w[u]);
    }
    if constexpr (EARLY_TW) {
#pragma unroll
      for (int u = 0; u < UNR; u++) tw_load_at(tid, u);
    }
    if (fused) {       // first stage pair next to the loads: the data of a wave starts computing when IT has arrived
#pragma unroll
      for (int q = 0; q < GQ; q++) {
        Fe<P> x0 = fe_unpack<P>(w[q]), x2 = fe_unpack<P>(w[q + GQ]), x1 = fe_unpack<P>(w[q + 2 * GQ]), x3 = fe_unpack<P>(w[q + 3 * GQ]);
        first_pair_regs<P, G>(x0, x1, x2, x3, tw_tile, tw_shoup, lgn);
        const int r = (tid >> lgc) + q * (G::NT >> lgc), c = tid & cmask;
        const int k0 = (int)(__brev((unsigned)r) >> (32 - (lgn - 2))) << 2;
        const int p0 = (k0 << lgc) | c, d1 = 1 << lgc;
        const int ph0 = G::phys(p0), ph1 = ph0 ^ G::phys(d1), ph2 = ph0 ^ G::phys(2 * d1);
        lds_store_ph<P, G>(lds, ph0, x0);
        lds_store_ph<P, G>(lds, ph1, x1);
        lds_store_ph<P, G>(lds, ph2, x2);
        lds_store_ph<P, G>(lds, ph1 ^ G::phys(2 * d1), x3);
      }
    } else
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int e = tid + u * G::NT;
      const int j1 = e >> lgc, c = e & cmask;
      const int k = (int)(__brev((unsigned)j1) >> (32 - lgn));
      lds_store<P, G>(lds, (k << lgc) | c, fe_unpack<P>(w[u]));
    }
  } else {
  if constexpr (EARLY_TW) {      // the coefficient vector is a fraction of the tile (the rest is the zero padding): the twiddles may as well go first
#pragma unroll
    for (int u = 0; u < UNR; u++) tw_load_at(tid, u);
  }
  for (int e = tid; e < tile_elems; e += G::NT) {
    const int j1 = e >> lgc, c = e & cmask;
    Fe<P> v;
    if constexpr (PRE) {
      // position inside transform `o` (the first pass spans a whole transform: lgn + lgM = log2 order); a batch stores
      // the coefficient vectors back to back, n_coef apart
      const size_t idx = ((size_t)j1 << lgM) + (ct << lgc) + c;
      if (idx < pre.n_coef) {
        const Fe<P> x = gload<P>(pre.coef, o * pre.n_coef + idx);
        const Fe<P> wc = gload<P>(pre.pre_col, (ct << lgc) + c), wr = gload<P>(pre.pre_row, (size_t)j1);
        v = fe_fit<P>(FeAsm<P>::mul(FeAsm<P>::mul(x, wc), wr));    // plain * Montgomery * Montgomery = plain
      } else {
        v = fe_zero<P>();
      }
    } else {
      v = gload<P>(in, base + ((size_t)j1 << lgM) + c);
    }
    const int k = (int)(__brev((unsigned)j1) >> (32 - lgn));
    lds_store<P, G>(lds, (k << lgc) | c, v);
  }
  }
  __syncthreads();
  if constexpr (MID_TW) {
#pragma unroll
    for (int u = 0; u < UNR; u++) tw_load_at(tid, u);
  }
  tile_stages<P, G>(lds, twl, lgn, lgc, (fused && !PRE) ? 3 - (lgn & 1) : 1, fused ? lgn - 2 : lgn, tw_shoup);
  {
    // The inter-pass twiddles of all the lane's elements, requested before the first product.  The epilogue's addresses are
    // recomputed from an OPAQUE copy of the lane id: hoisted above the stage loop they were 22 more registers live across it,
    // and the 1024-lane BN254 tile (128 VGPRs per lane) spilled exactly those (vgpr_spill_count 32 -> 0; tests/test_abi_load.py
    // keeps every hot kernel at zero).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if constexpr (!EARLY_TW && !MID_TW) {
#pragma unroll
      for (int u = 0; u < UNR; u++) tw_load_at(tid, u);
    }
    if (fused) {       // last stage pair on registers: the lane's group q is rows j1 + v 2^(lgn-2), exactly the elements it stores
#pragma unroll
      for (int q = 0; q < GQ; q++) {
        const int j1 = (tid >> lgc) + q * (G::NT >> lgc), c = tid & cmask;
        const int p0 = (j1 << lgc) | c, d1 = 1 << (lgn - 2 + lgc);
        Fe<P> x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = lds_load_ph<P, G>(lds, G::phys(p0) ^ G::phys(u * d1));
        radix4_wave<P, G>(x[0], x[1], x[2], x[3], twl, 1 << (lgn - 1), lgn, lgn - 1, j1);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const size_t off = ((size_t)(j1 + (u << (lgn - 2))) << lgM) + (ct << lgc) + c;
          gstore_stream<P>(out, (o << (lgn + lgM)) + off, inter_mul<P>(x[u], fe_unpack<P>(tw[q + GQ * u])), wt);
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int e = tid + u * G::NT;
      const int k = e >> lgc, c = e & cmask;
      const size_t off = ((size_t)k << lgM) + (ct << lgc) + c;
      const Fe<P> v = lds_load<P, G>(lds, (k << lgc) | c);
      gstore_stream<P>(out, (o << (lgn + lgM)) + off, inter_mul<P>(v, fe_unpack<P>(tw[u])), wt);
    }
  }
}

// Last pass: contiguous rows of 2^lgn, transposed scatter; optional final scale (n^-1 for a
// single-pass inverse), canonical output.  The grid may cover a BATCH of transforms stored back to back (row
// r = blockIdx * 2^lgr + rr belongs to transform r >> lg_rows): the product trees of mzk_poly.hip transform hundreds
// of small polynomials per launch.
template <class P, class G, bool WT>
__global__ __launch_bounds__(G::NT, G::WPE) void k_ntt_last(const u32* __restrict__ in, u32* __restrict__ out,
                                                        const u32* __restrict__ tw_tile, LevelInfo li, int lgn, int lgr,
                                                        int lg_rows, Words8 scale, int has_scale, size_t total_rows, int fuse, const u32* __restrict__ tw_shoup) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const int tid = threadIdx.x;
  const size_t p0 = (size_t)blockIdx.x << lgr;
  const int rmask = (1 << lgr) - 1, nmask = (1 << lgn) - 1;
  const int tile_elems = 1 << (lgn + lgr);
  const size_t rowmask = ((size_t)1 << lg_rows) - 1;
  const int logn = lg_rows + lgn;
  const u32* twl = G::TWG ? tw_tile : lds + P::L * G::TILE;
  stage_twiddles<P, G>(lds + P::L * G::TILE, tw_tile, lgn);
  constexpr int UNR = G::TILE / G::NT;
  constexpr int GQ = G::GQ;
  static_assert(UNR == 4 * GQ, "whole radix-4 groups per lane");
  constexpr bool wt = WT;
  const bool fused = fuse_edges<G>(lgn, lgr, fuse);      // full tile, even number of levels: first and last stage pair on registers
  auto row_base = [&](size_t r) -> size_t {        // first element of logical row r in the previous pass's layout
    size_t rem = r & rowmask, row = 0;
    for (int i = 0; i < li.nlev - 1; i++) {
      row = (row << li.lg[i]) | (rem & (((size_t)1 << li.lg[i]) - 1));
      rem >>= li.lg[i];
    }
    return ((r >> lg_rows) << logn) + (row << lgn);
  };
  if (fused) {
    // lane = (row rr, r): its loads j = r + u 2^(lgn-2) are one radix-4 group of the first stage pair (bit-reversed rows
    // 4 brev(r) + brev2(u)); consecutive lanes read consecutive elements of a row
    const int r4 = tid & ((1 << (lgn - 2)) - 1);
    u32 w[GQ][4][P::NW];
#pragma unroll
    for (int g = 0; g < GQ; g++) {           // group g of the lane: row rr + g (NT >> (lgn - 2)) of the tile
      const size_t r = p0 + (tid >> (lgn - 2)) + g * (G::NT >> (lgn - 2));
#pragma unroll
      for (int u = 0; u < 4; u++) {
#pragma unroll
        for (int q = 0; q < P::NW; q++) w[g][u][q] = 0;
        if (r < total_rows) gload_words_stream<P>(in, row_base(r) + r4 + ((size_t)u << (lgn - 2)), w[g][u]);
      }
    }
#pragma unroll
    for (int g = 0; g < GQ; g++) {
      const int rr = (tid >> (lgn - 2)) + g * (G::NT >> (lgn - 2));
      Fe<P> x0 = fe_unpack<P>(w[g][0]), x2 = fe_unpack<P>(w[g][1]), x1 = fe_unpack<P>(w[g][2]), x3 = fe_unpack<P>(w[g][3]);
      first_pair_regs<P, G>(x0, x1, x2, x3, tw_tile, tw_shoup, lgn);
      const int k0 = (int)(__brev((unsigned)r4) >> (32 - (lgn - 2))) << 2;
      const int q0 = (k0 << lgr) | rr, d1 = 1 << lgr;
      const int ph0 = G::phys(q0), ph1 = ph0 ^ G::phys(d1), ph2 = ph0 ^ G::phys(2 * d1);
      lds_store_ph<P, G>(lds, ph0, x0);
      lds_store_ph<P, G>(lds, ph1, x1);
      lds_store_ph<P, G>(lds, ph2, x2);
      lds_store_ph<P, G>(lds, ph1 ^ G::phys(2 * d1), x3);
    }
  } else
  for (int e0 = tid; e0 < tile_elems; e0 += UNR * G::NT) {       // one trip for a full tile: all loads first, then the LDS stores
    u32 w[UNR][P::NW];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int e = e0 + u * G::NT;
      const int rr = e >> lgn, j = e & nmask;
      const size_t r = p0 + rr;
#pragma unroll
      for (int q = 0; q < P::NW; q++) w[u][q] = 0;
      if (e < tile_elems && r < total_rows) gload_words_stream<P>(in, row_base(r) + j, w[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int e = e0 + u * G::NT;
      if (e >= tile_elems) break;
      const int rr = e >> lgn, j = e & nmask;
      const int k = (lgn == 0) ? 0 : (int)(__brev((unsigned)j) >> (32 - lgn));
      lds_store<P, G>(lds, (k << lgr) | rr, fe_unpack<P>(w[u]));
    }
  }
  __syncthreads();
  tile_stages<P, G>(lds, twl, lgn, lgr, fused ? 3 - (lgn & 1) : 1, fused ? lgn - 2 : lgn, tw_shoup);
  Fe<P> sc = fe_zero<P>();
  if (has_scale) sc = fe_unpack<P>(scale.w);
  if (fused) {         // last stage pair on registers: the group of lane (j1, rr) is rows j1 + u 2^(lgn-2), the elements it stores
    const int rr = tid & rmask;
    const size_t r = p0 + rr;
#pragma unroll
    for (int g = 0; g < GQ; g++) {
      const int j1 = (tid >> lgr) + g * (G::NT >> lgr);
      const int q0 = (j1 << lgr) | rr, d1 = 1 << (lgn - 2 + lgr);
      Fe<P> x[4];
#pragma unroll
      for (int u = 0; u < 4; u++) x[u] = lds_load_ph<P, G>(lds, G::phys(q0) ^ G::phys(u * d1));
      radix4_wave<P, G>(x[0], x[1], x[2], x[3], twl, 1 << (lgn - 1), lgn, lgn - 1, j1);
      if (r >= total_rows) continue;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        gstore_stream<P>(out, ((r >> lg_rows) << logn) + (r & rowmask) + ((size_t)(j1 + (u << (lgn - 2))) << lg_rows), final_reduce<P>(x[u], sc, has_scale), wt);
      }
    }
    return;
  }
  for (int e = tid; e < tile_elems; e += G::NT) {
    const int rr = e & rmask, k = e >> lgr;
    const size_t r = p0 + rr;
    if (r >= total_rows) continue;
    const Fe<P> v = lds_load<P, G>(lds, (k << lgr) | rr);
    gstore_stream<P>(out, ((r >> lg_rows) << logn) + (r & rowmask) + ((size_t)k << lg_rows), final_reduce<P>(v, sc, has_scale), wt);
  }
}

// `cols` transforms of W = 2^LGW points each, COLUMN-major: element i of transform c sits at [i * cols + c] (in and out).  One
// lane per column: W strided loads (consecutive lanes read consecutive elements), the radix-2 DIT stages in registers, W
// strided stores -- one trip over the data.  This is the step ACROSS the ranks of a transform sharded over several GPUs
// (myzkp_amd/sharded.py: after the first all-to-all a rank holds [rank a][its slice of t]); the transposes + row transforms
// it replaces were three trips.  tw: w^j, j < W/2, Montgomery (a plan's in-tile table); has_scale: W^-1 of the inverse.
template <class P, int LGW>
__global__ __launch_bounds__(256) void k_ntt_columns(const u32* __restrict__ in, u32* __restrict__ out, size_t cols,
                                                      const u32* __restrict__ tw, Words8 scale, int has_scale) {
  constexpr int W = 1 << LGW;
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  Fe<P> x[W];
#pragma unroll
  for (int i = 0; i < W; i++) x[(int)(__brev((unsigned)i) >> (32 - LGW))] = gload<P>(in, (size_t)i * cols + c);
#pragma unroll
  for (int s = 1; s <= LGW; s++) {
    const int half = 1 << (s - 1);
#pragma unroll
    for (int b = 0; b < W / 2; b++) {
      const int j = b & (half - 1), g = b >> (s - 1);
      const int lo = (g << s) | j, hi = lo + half;
      Fe<P> t = x[hi];
      if (j != 0) t = FeAsm<P>::mul(t, gload<P>(tw, (size_t)(j << (LGW - s))));
      else if (s != 1) t = fe_weak_reduce<P>(t);
      x[hi] = fe_sub_carry<P, 8>(x[lo], t);
      x[lo] = fe_add_carry<P>(x[lo], t);
    }
  }
  Fe<P> sc;
  if (has_scale) sc = fe_unpack<P>(scale.w);
#pragma unroll
  for (int k = 0; k < W; k++) {
    Fe<P> v = x[k];
    if (has_scale) v = FeAsm<P>::mul(v, sc);
    gstore<P>(out, (size_t)k * cols + c, fe_reduce<P>(v));
  }
}

// out[c * rows + r] = in[r * cols + c] over whole elements (NW words each): the interleave that turns the landing buffer of
// the last exchange of a sharded transform, [source rank][k2'], into natural order k2' * W + rank.
template <int NW>
__global__ void k_transpose_elems(const u32* __restrict__ in, u32* __restrict__ out, size_t rows, size_t cols) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // output element index: c * rows + r
  if (i >= rows * cols) return;
  const size_t c = i / rows, r = i - c * rows;
  const uint4* src = reinterpret_cast<const uint4*>(in + (r * cols + c) * NW);
  uint4* dst = reinterpret_cast<uint4*>(out + i * NW);
#pragma unroll
  for (int q = 0; q < NW / 4; q++) dst[q] = src[q];
}

// ---- table generation ------------------------------------------------------------------------------
constexpr int GEN_CHUNK = 16;
// out[j] = g^j (Montgomery, canonical), j < count, g = root^emul
template <class P>
__global__ void k_gen_pow_table(Words8 root_plain, uint64_t emul, size_t count, u32* __restrict__ out) {
  const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j0 = chunk * GEN_CHUNK;
  if (j0 >= count) return;
  Fe<P> g = fe_pow_u64<P>(fe_to_mont<P>(fe_unpack<P>(root_plain.w)), emul);
  Fe<P> cur = fe_pow_u64<P>(g, j0);
  for (int i = 0; i < GEN_CHUNK && j0 + i < count; i++) {
    gstore<P>(out, j0 + i, fe_reduce<P>(cur));
    cur = fe_mul<P>(cur, g);
  }
}
// out[k * M + j'] = scale * g^(j' k), g = root^emul a primitive (n_t M)-th root; k < 2^lgn, j' < 2^lgM
template <class P>
__global__ void k_gen_inter_table(Words8 root_plain, uint64_t emul, int lgn, int lgM, Words8 scale_plain,
                                  u32* __restrict__ out) {
  const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t M = (size_t)1 << lgM;
  const size_t clen = M < GEN_CHUNK ? M : GEN_CHUNK;
  const size_t chunks_per_row = M / clen;
  const size_t k = chunk / chunks_per_row, j0 = (chunk % chunks_per_row) * clen;
  if (k >= ((size_t)1 << lgn)) return;
  Fe<P> g = fe_pow_u64<P>(fe_to_mont<P>(fe_unpack<P>(root_plain.w)), emul);
  Fe<P> gk = fe_pow_u64<P>(g, k);
  Fe<P> cur = fe_mul<P>(fe_pow_u64<P>(gk, j0), fe_to_mont<P>(fe_unpack<P>(scale_plain.w)));
  for (size_t i = 0; i < clen; i++) {
    gstore<P>(out, k * M + j0 + i, fe_reduce<P>(cur));
    cur = fe_mul<P>(cur, gk);
  }
}
// Polynomial::scale (polynomial.rs:167-174) + zero padding (ntt.rs:264-267): out[i] = coef[i] * offset^i
// for i < n_coef, 0 for n_coef <= i < order.
template <class P>
__global__ void k_coset_scale_pad(const u32* __restrict__ coef, size_t n_coef, Words8 offset_plain,
                                  u32* __restrict__ out, size_t order, size_t chunks_per, size_t batch) {
  const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t o = chunk / chunks_per;                 // transform of the batch (vectors back to back: n_coef in, order out)
  if (o >= batch) return;
  const size_t j0 = (chunk - o * chunks_per) * GEN_CHUNK;
  if (j0 >= order) return;
  coef += o * n_coef * P::NW;
  out += o * order * P::NW;
  if (j0 >= n_coef) {
    for (int i = 0; i < GEN_CHUNK && j0 + i < order; i++) gstore<P>(out, j0 + i, fe_zero<P>());
    return;
  }
  Fe<P> g = fe_to_mont<P>(fe_unpack<P>(offset_plain.w));
  Fe<P> cur = fe_pow_u64<P>(g, j0);
  for (int i = 0; i < GEN_CHUNK && j0 + i < order; i++) {
    if (j0 + i < n_coef) {
      Fe<P> c = gload<P>(coef, j0 + i);
      gstore<P>(out, j0 + i, fe_reduce<P>(FeAsm<P>::mul(c, cur)));
      cur = FeAsm<P>::mul(cur, g);
    } else {
      gstore<P>(out, j0 + i, fe_zero<P>());
    }
  }
}
// Polynomial::scale (polynomial.rs:167-174) with a leading constant: out[i] = lead * in[i] * ratio^i, plain in and out,
// in place allowed.  The twiddle step of the sharded transforms (myzkp_amd/sharded.py) and their n^-1.
template <class P>
__global__ void k_poly_scale(const u32* __restrict__ in, size_t n, Words8 ratio_plain, Words8 lead_plain, u32* __restrict__ out) {
  const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j0 = chunk * GEN_CHUNK;
  if (j0 >= n) return;
  const Fe<P> g = fe_to_mont<P>(fe_unpack<P>(ratio_plain.w));
  Fe<P> cur = FeAsm<P>::mul(fe_pow_u64<P>(g, j0), fe_to_mont<P>(fe_unpack<P>(lead_plain.w)));
  for (int i = 0; i < GEN_CHUNK && j0 + i < n; i++) {
    const Fe<P> c = gload<P>(in, j0 + i);
    gstore<P>(out, j0 + i, fe_reduce<P>(FeAsm<P>::mul(c, cur)));
    cur = FeAsm<P>::mul(cur, g);
  }
}
// out[i] = a[i] * b[i] (plain domain in and out; Hadamard step of the polynomial products)
template <class P>
__global__ void k_pointwise_mul(const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<P> x = gload<P>(a, i), y = gload<P>(b, i);
  Fe<P> t = FeAsm<P>::mul(FeAsm<P>::mul(x, y), fe_r2<P>());
  gstore<P>(out, i, fe_reduce<P>(t));
}

// ---- plans -------------------------------------------------------------------------------------------
struct NttPlan {
  int fid = -1;
  unsigned logn = 0;
  bool inverse = false;
  bool large = false;                  // tile geometry the level split and the tables were built for (large_geo)
  uint64_t root[4] = {0, 0, 0, 0};     // forward root as passed by the caller
  uint64_t scale[4] = {0, 0, 0, 0};    // extra plain scale folded into the tables (1 if none)
  LevelInfo li{};
  u32* tw_tile[4] = {nullptr, nullptr, nullptr, nullptr};
  u32* tw_inter[3] = {nullptr, nullptr, nullptr};
  u32* tw_shoup[4] = {nullptr, nullptr, nullptr, nullptr};   // Fr: per level, the in-tile twiddles again as (plain w, floor(w 2^261 / p)) limb entries
  Words8 last_scale{};                 // single-pass inverse: n^-1 (* scale), Montgomery
  int has_last_scale = 0;
  uint64_t stamp = 0;
};
static std::vector<NttPlan*> g_plans_all[MZK_MAX_CTX];      // per context (device)
#define g_plans g_plans_all[ctx().index]
static uint64_t g_stamp = 0;
constexpr size_t MAX_PLANS = 96;      // the polynomial trees of mzk_poly.hip use every size from 2^7 up, both directions

static void free_plan(NttPlan* p) {
  for (auto& t : p->tw_tile) if (t) (void)hipFree(t);
  for (auto& t : p->tw_inter) if (t) (void)hipFree(t);
  for (auto& t : p->tw_shoup) if (t) (void)hipFree(t);
  delete p;
}
// offset-power tables of the fused coset LDE, per (context, field); see coset_lde_dev_impl
static struct { uint64_t off[4]; unsigned logn; int lgn0; uint64_t gen; bool valid; hipEvent_t ready; } g_lde_cache[MZK_MAX_CTX][2] = {};
void ntt_release_plans() {
  for (auto* p : g_plans) free_plan(p);
  g_plans.clear();
  for (auto& ce : g_lde_cache[ctx().index]) {      // mzk_shutdown walks the contexts: drop this one's entries and their events
    if (ce.ready) (void)hipEventDestroy(ce.ready);
    memset(&ce, 0, sizeof ce);
  }
}

static LevelInfo choose_levels(unsigned logn, bool large) {
  LevelInfo li{};
  if (logn <= TILE_LOG) {
    li.nlev = 1;
    li.lg[0] = (int)logn;
    return li;
  }
  const int maxlv = large ? GeoL::MAXLV : MAX_LEVEL_LOG;
  int k = (int)((logn + maxlv - 1) / maxlv);
  li.nlev = k;
  int base = (int)logn / k, extra = (int)logn % k;
  for (int i = 0; i < k; i++) li.lg[i] = base + (i < extra ? 1 : 0);
  return li;
}

static void to_words(const uint64_t* limbs, int nl, Words8* w) {
  memset(w, 0, sizeof *w);
  for (int i = 0; i < nl; i++) {
    w->w[2 * i] = (u32)limbs[i];
    w->w[2 * i + 1] = (u32)(limbs[i] >> 32);
  }
}

// Shoup entries of the in-tile twiddles w^j, j < cnt, w = root^emul: limbs (29 bits) of w^j at words [0, 9), of
// floor(w^j 2^261 / p) at [16, 25) of a 32-word entry.  Host arithmetic (a few hundred 256-bit products and divisions per plan).
static int build_shoup_table(const HostField* hf, const uint64_t* root, uint64_t emul, size_t cnt, u32** out, hipStream_t s) {
  std::vector<u32> tab(cnt * SHOUP_ENTRY_WORDS, 0u);
  uint64_t w[4], cur[4] = {1, 0, 0, 0};
  h_powmod_u64(hf, w, root, emul);
  auto limbs_of = [](const uint64_t* v5, u32* dst) {      // 9 limbs of 29 bits from a 5 x 64-bit number
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, k = bit >> 6, sft = bit & 63;
      uint64_t x = v5[k] >> sft;
      if (sft > 35 && k + 1 < 5) x |= v5[k + 1] << (64 - sft);
      dst[i] = (u32)(x & 0x1fffffffu);
    }
  };
  for (size_t j = 0; j < cnt; j++) {
    uint64_t v5[5] = {cur[0], cur[1], cur[2], cur[3], 0};
    limbs_of(v5, &tab[j * SHOUP_ENTRY_WORDS]);
    // q = floor(cur 2^261 / p): restoring division, one quotient bit per step (cur < p < 2^254, so the remainder stays below 2^255)
    uint64_t rem[4] = {cur[0], cur[1], cur[2], cur[3]}, q[5] = {0, 0, 0, 0, 0};
    for (int step = 0; step < 261; step++) {
      for (int i = 4; i > 0; i--) q[i] = (q[i] << 1) | (q[i - 1] >> 63);
      q[0] <<= 1;
      for (int i = 3; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
      rem[0] <<= 1;
      bool ge = true;
      for (int i = 3; i >= 0; i--) if (rem[i] != hf->p[i]) { ge = rem[i] > hf->p[i]; break; }
      if (ge) {
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; i++) {
          const unsigned __int128 d = (unsigned __int128)rem[i] - hf->p[i] - (uint64_t)br;
          rem[i] = (uint64_t)d;
          br = (d >> 64) & 1;
        }
        q[0] |= 1;
      }
    }
    limbs_of(q, &tab[j * SHOUP_ENTRY_WORDS + 16]);
    h_mulmod(hf, cur, cur, w);
  }
  MZK_HIP(hipMalloc((void**)out, tab.size() * sizeof(u32)));
  MZK_HIP(hipMemcpyAsync(*out, tab.data(), tab.size() * sizeof(u32), hipMemcpyHostToDevice, s));
  MZK_HIP(hipStreamSynchronize(s));          // `tab` is a local
  return MZK_OK;
}

template <class P>
static int build_tables(NttPlan* pl, const uint64_t* eff_root, const uint64_t* fold_scale, hipStream_t s) {
  const HostField* hf = host_field(pl->fid);
  Words8 rootw, scalew;
  to_words(eff_root, hf->nl, &rootw);
  to_words(fold_scale, hf->nl, &scalew);
  uint64_t one[4] = {1, 0, 0, 0};
  Words8 onew;
  to_words(one, hf->nl, &onew);
  const LevelInfo& li = pl->li;
  const size_t esz = sizeof(u32) * P::NW;
  int lg_after = (int)pl->logn;  // log of (n_t * M_t) while walking levels
  for (int t = 0; t < li.nlev; t++) {
    const int lgn = li.lg[t];
    const int lgM = lg_after - lgn;
    // in-tile twiddles: w_{n_t}^j = root^(j * n / n_t), j < n_t / 2
    if (lgn >= 2) {
      size_t cnt = (size_t)1 << (lgn - 1);
      MZK_HIP(hipMalloc((void**)&pl->tw_tile[t], cnt * esz));
      unsigned blocks = (unsigned)((cnt + GEN_CHUNK * 64 - 1) / (GEN_CHUNK * 64));
      hipLaunchKernelGGL((k_gen_pow_table<P>), dim3(blocks), dim3(64), 0, s, rootw, (uint64_t)1 << (pl->logn - lgn), cnt,
                         pl->tw_tile[t]);
      if (HasShoup<P>::value) MZK_TRY(build_shoup_table(hf, eff_root, (uint64_t)1 << (pl->logn - lgn), cnt, &pl->tw_shoup[t], s));
    }
    if (t < li.nlev - 1) {
      // inter-pass twiddles for the block of size n_t * M_t: primitive root = root^(n / (n_t M_t))
      size_t cnt = (size_t)1 << lg_after;
      MZK_HIP(hipMalloc((void**)&pl->tw_inter[t], cnt * esz));
      size_t clen = ((size_t)1 << lgM) < (size_t)GEN_CHUNK ? ((size_t)1 << lgM) : (size_t)GEN_CHUNK;
      size_t chunks = cnt / clen;
      unsigned blocks = (unsigned)((chunks + 255) / 256);
      hipLaunchKernelGGL((k_gen_inter_table<P>), dim3(blocks), dim3(256), 0, s, rootw,
                         (uint64_t)1 << (pl->logn - lg_after), lgn, lgM, (t == 0) ? scalew : onew, pl->tw_inter[t]);
    }
    lg_after = lgM;
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

static int get_plan(int fid, unsigned logn, bool inverse, const uint64_t* root, const uint64_t* extra_scale,
                    hipStream_t s, NttPlan** out, size_t batch = 1) {
  const bool large = large_geo(fid, logn, batch);
  const HostField* hf = host_field(fid);
  uint64_t one[4] = {1, 0, 0, 0};
  const uint64_t* sc = extra_scale ? extra_scale : one;
  for (auto* p : g_plans) {
    if (p->fid == fid && p->logn == logn && p->inverse == inverse && p->large == large && !memcmp(p->root, root, 8 * hf->nl) &&
        !memcmp(p->scale, sc, 8 * hf->nl)) {
      p->stamp = ++g_stamp;
      *out = p;
      return MZK_OK;
    }
  }
  // reference assertions, ntt.rs:15-22
  const uint64_t n = (uint64_t)1 << logn;
  uint64_t t[4];
  h_powmod_u64(hf, t, root, n);
  if (!h_is_one(hf, t)) { set_error("primitive root must be nth root of unity, where n is len(values)"); return MZK_E_ROOT_ORDER; }
  h_powmod_u64(hf, t, root, n / 2);
  if (h_is_one(hf, t)) { set_error("primitive root is not primitive nth root of unity, where n is len(values)"); return MZK_E_ROOT_PRIM; }

  NttPlan* pl = new NttPlan();
  pl->fid = fid; pl->logn = logn; pl->inverse = inverse; pl->large = large;
  memcpy(pl->root, root, 8 * hf->nl);
  memcpy(pl->scale, sc, 8 * hf->nl);
  pl->li = choose_levels(logn, large);
  pl->stamp = ++g_stamp;
  uint64_t eff_root[4] = {0, 0, 0, 0}, fold[4] = {0, 0, 0, 0};
  memcpy(fold, sc, 8 * hf->nl);
  if (inverse) {
    h_powmod_u64(hf, eff_root, root, n - 1);  // root^-1 = root^(n-1)          (ntt.rs:59)
    uint64_t ninv[4];
    h_ninv_pow2(hf, logn, ninv);              // F::from_value(n).inverse()    (ntt.rs:58)
    h_mulmod(hf, fold, fold, ninv);
  } else {
    memcpy(eff_root, root, 8 * hf->nl);
  }
  bool fold_is_one = h_is_one(hf, fold);
  if (pl->li.nlev == 1 && !fold_is_one) {
    // single pass: no inter-pass table to fold into -> explicit scale in the last pass (Montgomery form
    // computed on the host: fold * R mod p via mulmod with R mod p)
    uint64_t r_mod_p[4] = {0, 0, 0, 0}, two[4] = {2, 0, 0, 0};
    // R = 2^(29 L): square-and-multiply on the host
    h_powmod_u64(hf, r_mod_p, two, (uint64_t)(29 * (fid == MZK_FIELD_M128 ? 5 : 9)));
    uint64_t m[4];
    h_mulmod(hf, m, fold, r_mod_p);
    to_words(m, hf->nl, &pl->last_scale);
    pl->has_last_scale = 1;
  }
  int rc = (fid == MZK_FIELD_M128) ? build_tables<M128Params>(pl, eff_root, fold, s)
                                   : build_tables<FrParams>(pl, eff_root, fold, s);
  if (rc != MZK_OK) { free_plan(pl); return rc; }
  if (g_plans.size() >= MAX_PLANS) {
    size_t victim = 0;
    for (size_t i = 1; i < g_plans.size(); i++) if (g_plans[i]->stamp < g_plans[victim]->stamp) victim = i;
    MZK_HIP(hipStreamSynchronize(s));
    free_plan(g_plans[victim]);
    g_plans.erase(g_plans.begin() + victim);
  }
  g_plans.push_back(pl);
  *out = pl;
  return MZK_OK;
}

template <class P, class G>
static int run_plan_geo(const NttPlan* pl, const u32* d_in, u32* d_out, hipStream_t s, const PreArgs* pre, size_t batch) {
  const LevelInfo& li = pl->li;
  const unsigned logn = pl->logn;

  ProfScope whole(s, MZK_PH_NTT_TOTAL);
  static const int fuse = tune_int("MZK_NTT_FUSE_EDGES", 1);     // 0: A/B (tools/timing/time_ntt.py)
  static const int shoup = tune_int("MZK_NTT_SHOUP", 1);
  // write-through stores (stream_store16) for multi-pass transforms whose data is between 2^WT_LO and 2^WT_HI bytes; mask bit 0:
  // the strided passes, bit 1: the last pass
  // (same-box map, profiles/round5_ntt_write_through_ab.txt: Fr 2^17..2^19 -1..-13 %, M128 2^17..2^22 -2..-16 %; below 2 MiB and
  // from 32 MiB (Fr) / 128 MiB (M128) up it gains nothing or costs 3-9 %: the working set no longer sits in the Infinity Cache)
  static const int wt_lo = tune_int("MZK_NTT_WT_LO", 21), wt_hi = tune_int("MZK_NTT_WT_HI", P::NW == 4 ? 26 : 24), wt_mask = tune_int("MZK_NTT_WT_MASK", 3);
  const size_t bytes = (batch << logn) * sizeof(u32) * P::NW;
  const bool wt_on = li.nlev > 1 && bytes >= ((size_t)1 << wt_lo) && bytes <= ((size_t)1 << wt_hi);
  const bool wt_s = wt_on && (wt_mask & 1), wt_l = wt_on && (wt_mask & 2);
  if constexpr (G::TL != TILE_LOG) {        // tiles above 64 KiB of LDS need the attribute, once per context and instantiation
    bool& done = ctx().attr_done[P::NW == 4 ? (G::TWG ? ATTR_NTT_LARGE_M128_2WG : ATTR_NTT_LARGE_M128) : ATTR_NTT_LARGE_FR];
    if (!done) {
      const void* fns[] = {(const void*)k_ntt_strided<P, true, G, false>, (const void*)k_ntt_strided<P, false, G, false>, (const void*)k_ntt_last<P, G, false>,
                           (const void*)k_ntt_strided<P, true, G, true>,  (const void*)k_ntt_strided<P, false, G, true>,  (const void*)k_ntt_last<P, G, true>};
      for (const void* f : fns) MZK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done = true;
    }
  }
  const u32* src = d_in;
  u32* tmp = nullptr;
  if (li.nlev > 1) MZK_TRY(ws_get(WS_NTT_TMP, (batch << logn) * sizeof(u32) * P::NW, (void**)&tmp));
  int lg_after = (int)logn;
  if constexpr (G::MAXLV < G::TL || G::TL != TILE_LOG)      // (GeoS1 serves single-pass plans only: no strided kernels of its own)
  for (int t = 0; t < li.nlev - 1; t++) {
    const int lgn = li.lg[t], lgM = lg_after - lgn;
    const int lgc = G::TL - lgn;
    const unsigned blocks = (unsigned)(batch << (logn - G::TL));      // `o` in the kernel runs over the batch too
    {
      ProfScope ps(s, MZK_PH_NTT_PASS0 + t);
      const bool with_pre = t == 0 && pre;
      auto* k = with_pre ? (wt_s ? k_ntt_strided<P, true, G, true> : k_ntt_strided<P, true, G, false>)
                         : (wt_s ? k_ntt_strided<P, false, G, true> : k_ntt_strided<P, false, G, false>);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(G::NT), G::template lds_bytes<P>(lgn), s, src, tmp, pl->tw_tile[t], pl->tw_inter[t], lgn, lgM, lgc,
                         with_pre ? *pre : PreArgs{nullptr, 0, nullptr, nullptr}, fuse, shoup ? pl->tw_shoup[t] : nullptr);
    }
    src = tmp;
    lg_after = lgM;
  }
  {
    const int lgn = li.lg[li.nlev - 1];
    const int lg_rows = (int)logn - lgn;
    const size_t total_rows = batch << lg_rows;
    int lgr = G::TL - lgn;
    while (lgr > 0 && ((size_t)1 << lgr) > total_rows) lgr--;
    const unsigned blocks = (unsigned)((total_rows + ((size_t)1 << lgr) - 1) >> lgr);
    ProfScope ps(s, MZK_PH_NTT_PASS0 + li.nlev - 1);
    hipLaunchKernelGGL((wt_l ? k_ntt_last<P, G, true> : k_ntt_last<P, G, false>), dim3(blocks), dim3(G::NT), G::template lds_bytes<P>(lgn), s, src, d_out,
                       pl->tw_tile[li.nlev - 1], li, lgn, lgr, lg_rows, pl->last_scale, pl->has_last_scale, total_rows, fuse,
                       shoup ? pl->tw_shoup[li.nlev - 1] : nullptr);
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
template <class P>
static int run_plan(const NttPlan* pl, const u32* d_in, u32* d_out, hipStream_t s, const PreArgs* pre = nullptr, size_t batch = 1) {
  // M128 large tiles: two 512-lane workgroups per CU (GeoM) once there are at least two tiles per CU -- 2^25 and up: 1.69 instead of
  // 1.83 - 1.85 ms at 2^25 (same box A/B, profiles/round4_ntt_m128_two_workgroups_ab.txt).  A 2^20 transform is exactly 256 tiles, ONE per CU whatever the
  // workgroup size, and the 512-lane form only halves the lanes working on it (0.057 against 0.047 ms): it keeps round 3's
  // single 1024-lane workgroup.  Tuning build: MZK_NTT_M128_TWO_WG = smallest log2 size on GeoM (99 = never).
  static const int two_wg_from = tune_int("MZK_NTT_M128_TWO_WG", 21);
  if (pl->large && P::NW == 4 && (int)pl->logn >= two_wg_from) return run_plan_geo<P, typename LargeGeo<P>::type>(pl, d_in, d_out, s, pre, batch);
  if (pl->large) return run_plan_geo<P, GeoL>(pl, d_in, d_out, s, pre, batch);
  if (pl->li.nlev == 1 && pl->li.lg[0] > GeoS::MAXLV) return run_plan_geo<P, GeoS1>(pl, d_in, d_out, s, pre, batch);
  return run_plan_geo<P, GeoS>(pl, d_in, d_out, s, pre, batch);
}

static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static unsigned ilog2(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }

// extra_scale_host: optional plain constant multiplied into every output (used by the polynomial
// products to fold constants); nullptr = 1.
int ntt_dev_impl(int fid, const uint64_t* root_host, const void* d_in, void* d_out, size_t n, int inverse,
                 const uint64_t* extra_scale_host, hipStream_t s) {
  if (fid != MZK_FIELD_FR && fid != MZK_FIELD_M128) { set_error("ntt: field id %d has no NTT on this path", fid); return MZK_E_ARG; }
  if (n == 0) return MZK_OK;  // empty Vec in, empty Vec out (ntt.rs:12-14 `len <= 1`; len-1 underflow aside)
  if (!is_pow2(n)) { set_error("cannot compute ntt of non-power-of-two sequence"); return MZK_E_NOT_POW2; }
  if (!d_in || !d_out || (!root_host && n > 1)) { set_error("ntt: null pointer"); return MZK_E_ARG; }
  const size_t esz = field_bytes(fid);
  if (n == 1) {  // ntt.rs:12-14 / :54-56: returned unchanged
    if (d_in != d_out) MZK_HIP(hipMemcpyAsync(d_out, d_in, esz, hipMemcpyDeviceToDevice, s));
    return MZK_OK;
  }
  const HostField* hf = host_field(fid);
  if (!h_is_canonical(hf, root_host)) { set_error("ntt: root not canonical"); return MZK_E_RANGE; }
  if (ilog2(n) > 32) { set_error("ntt: n too large"); return MZK_E_ARG; }
  NttPlan* pl = nullptr;
  MZK_TRY(get_plan(fid, ilog2(n), inverse != 0, root_host, extra_scale_host, s, &pl));
  if (fid == MZK_FIELD_M128) return run_plan<M128Params>(pl, (const u32*)d_in, (u32*)d_out, s);
  return run_plan<FrParams>(pl, (const u32*)d_in, (u32*)d_out, s);
}

// `batch` transforms of n points each, stored back to back (in place allowed); same root for all
int ntt_batch_dev_impl(int fid, const uint64_t* root_host, const void* d_in, void* d_out, size_t n, size_t batch, int inverse, hipStream_t s) {
  if (batch == 0 || n == 0) return MZK_OK;
  if (batch == 1) return ntt_dev_impl(fid, root_host, d_in, d_out, n, inverse, nullptr, s);
  if (fid != MZK_FIELD_FR && fid != MZK_FIELD_M128) { set_error("ntt: field id %d has no NTT on this path", fid); return MZK_E_ARG; }
  if (!is_pow2(n)) { set_error("cannot compute ntt of non-power-of-two sequence"); return MZK_E_NOT_POW2; }
  if (!d_in || !d_out || (!root_host && n > 1)) { set_error("ntt: null pointer"); return MZK_E_ARG; }
  if (n > 1 && !h_is_canonical(host_field(fid), root_host)) { set_error("ntt: root not canonical"); return MZK_E_RANGE; }
  if (ilog2(n) > 32 || batch > ((size_t)1 << 40) / n) { set_error("ntt: batch too large"); return MZK_E_ARG; }
  if (n == 1) {
    if (d_in != d_out) MZK_HIP(hipMemcpyAsync(d_out, d_in, batch * field_bytes(fid), hipMemcpyDeviceToDevice, s));
    return MZK_OK;
  }
  NttPlan* pl = nullptr;
  MZK_TRY(get_plan(fid, ilog2(n), inverse != 0, root_host, nullptr, s, &pl, batch));
  if (fid == MZK_FIELD_M128) return run_plan<M128Params>(pl, (const u32*)d_in, (u32*)d_out, s, nullptr, batch);
  return run_plan<FrParams>(pl, (const u32*)d_in, (u32*)d_out, s, nullptr, batch);
}

template <class P, int LGW>
static void launch_columns(const NttPlan* pl, const void* d_in, void* d_out, size_t cols, hipStream_t s) {
  hipLaunchKernelGGL((k_ntt_columns<P, LGW>), dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, (const u32*)d_in, (u32*)d_out, cols,
                     (const u32*)pl->tw_tile[0], pl->last_scale, pl->has_last_scale ? 1 : 0);
}
template <class P>
static int columns_dispatch(const NttPlan* pl, unsigned lgw, const void* d_in, void* d_out, size_t cols, hipStream_t s) {
  switch (lgw) {
    case 1: launch_columns<P, 1>(pl, d_in, d_out, cols, s); break;
    case 2: launch_columns<P, 2>(pl, d_in, d_out, cols, s); break;
    case 3: launch_columns<P, 3>(pl, d_in, d_out, cols, s); break;
    case 4: launch_columns<P, 4>(pl, d_in, d_out, cols, s); break;
    default: set_error("ntt_columns: 2..16 points per transform"); return MZK_E_ARG;
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
// cols transforms of n_points (2, 4, 8 or 16) each, column-major; the same root checks and inverse convention as ntt_dev_impl
int ntt_columns_dev_impl(int fid, const uint64_t* root_host, const void* d_in, void* d_out, size_t n_points, size_t cols, int inverse, hipStream_t s) {
  if (fid != MZK_FIELD_FR && fid != MZK_FIELD_M128) { set_error("ntt_columns: field id %d has no NTT on this path", fid); return MZK_E_ARG; }
  if (cols == 0 || n_points == 0) return MZK_OK;
  if (!is_pow2(n_points)) { set_error("cannot compute ntt of non-power-of-two sequence"); return MZK_E_NOT_POW2; }
  if (n_points < 2 || n_points > 16) { set_error("ntt_columns: 2..16 points per transform"); return MZK_E_ARG; }
  if (!d_in || !d_out || !root_host) { set_error("ntt_columns: null pointer"); return MZK_E_ARG; }
  if (d_in == d_out) { set_error("ntt_columns: in place is not supported"); return MZK_E_ARG; }
  if (!h_is_canonical(host_field(fid), root_host)) { set_error("ntt: root not canonical"); return MZK_E_RANGE; }
  NttPlan* pl = nullptr;
  MZK_TRY(get_plan(fid, ilog2(n_points), inverse != 0, root_host, nullptr, s, &pl));
  if (fid == MZK_FIELD_M128) return columns_dispatch<M128Params>(pl, ilog2(n_points), d_in, d_out, cols, s);
  return columns_dispatch<FrParams>(pl, ilog2(n_points), d_in, d_out, cols, s);
}

int transpose_elems_dev_impl(int fid, const void* d_in, void* d_out, size_t rows, size_t cols, hipStream_t s) {
  const size_t total = rows * cols;
  if (total == 0) return MZK_OK;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (fid == MZK_FIELD_M128) hipLaunchKernelGGL((k_transpose_elems<4>), dim3(blocks), dim3(256), 0, s, (const u32*)d_in, (u32*)d_out, rows, cols);
  else hipLaunchKernelGGL((k_transpose_elems<8>), dim3(blocks), dim3(256), 0, s, (const u32*)d_in, (u32*)d_out, rows, cols);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

int poly_scale_dev_impl(int fid, const void* d_in, size_t n, const uint64_t* ratio_host, const uint64_t* lead_host, void* d_out, hipStream_t s) {
  if (fid != MZK_FIELD_FR && fid != MZK_FIELD_M128) { set_error("poly_scale: bad field id %d", fid); return MZK_E_ARG; }
  if (n == 0) return MZK_OK;
  if (!d_in || !d_out || !ratio_host) { set_error("poly_scale: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(fid);
  if (!h_is_canonical(hf, ratio_host) || (lead_host && !h_is_canonical(hf, lead_host))) { set_error("poly_scale: parameter not canonical"); return MZK_E_RANGE; }
  const uint64_t one[4] = {1, 0, 0, 0};
  Words8 rw, lw;
  to_words(ratio_host, hf->nl, &rw);
  to_words(lead_host ? lead_host : one, hf->nl, &lw);
  const size_t chunks = (n + GEN_CHUNK - 1) / GEN_CHUNK;
  const unsigned blocks = (unsigned)((chunks + 255) / 256);
  if (fid == MZK_FIELD_M128)
    hipLaunchKernelGGL((k_poly_scale<M128Params>), dim3(blocks), dim3(256), 0, s, (const u32*)d_in, n, rw, lw, (u32*)d_out);
  else
    hipLaunchKernelGGL((k_poly_scale<FrParams>), dim3(blocks), dim3(256), 0, s, (const u32*)d_in, n, rw, lw, (u32*)d_out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

int coset_lde_dev_impl(int fid, const void* d_coef, size_t n_coef, const uint64_t* offset_host,
                       const uint64_t* generator_host, void* d_out, size_t order, hipStream_t s, size_t batch) {
  if (batch == 0) return MZK_OK;
  if (fid != MZK_FIELD_FR && fid != MZK_FIELD_M128) { set_error("coset_lde: bad field id %d", fid); return MZK_E_ARG; }
  if (n_coef > order) { set_error("attempt to subtract with overflow (order - polynomial.coef.len())"); return MZK_E_LENGTH; }
  if (order == 0) return MZK_OK;
  if (!is_pow2(order)) { set_error("cannot compute ntt of non-power-of-two sequence"); return MZK_E_NOT_POW2; }
  if (!d_out || (!d_coef && n_coef) || !offset_host || !generator_host) { set_error("coset_lde: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(fid);
  if (!h_is_canonical(hf, offset_host) || !h_is_canonical(hf, generator_host)) { set_error("coset_lde: parameter not canonical"); return MZK_E_RANGE; }
  Words8 offw;
  to_words(offset_host, hf->nl, &offw);
  const unsigned logn = ilog2(order);
  if (logn > 32 || batch > ((size_t)1 << 40) / order) { set_error("coset_lde: order * batch too large"); return MZK_E_ARG; }
  if (order > 1 && choose_levels(logn, large_geo(fid, logn, batch)).nlev > 1) {
    // multi-pass transform: Polynomial::scale + padding fused into the first pass (PreArgs)
    if (!h_is_canonical(hf, generator_host)) { set_error("coset_lde: parameter not canonical"); return MZK_E_RANGE; }
    NttPlan* pl = nullptr;
    MZK_TRY(get_plan(fid, logn, false, generator_host, nullptr, s, &pl, batch));
    const int lgn0 = pl->li.lg[0], lgM0 = (int)logn - lgn0;
    const size_t nrow = (size_t)1 << lgn0, ncol = (size_t)1 << lgM0;
    // offset^(j M) and offset^col tables: like the plan's twiddles they depend only on (field, offset, size) -- a STARK
    // prover evaluates every polynomial on ONE coset -- so the last pair per field is kept (workspace generation and
    // stream order checked like the fixed-base tables in mzk_kzg.hip).
    auto& ce = g_lde_cache[ctx().index][fid == MZK_FIELD_M128 ? 1 : 0];
    void* tabs = nullptr;
    MZK_TRY(ws_get(fid == MZK_FIELD_M128 ? WS_NTT_PRE_M128 : WS_NTT_PRE, (nrow + ncol) * field_bytes(fid), &tabs));
    u32* pre_row = (u32*)tabs;
    u32* pre_col = pre_row + nrow * field_words(fid);
    if (!ce.ready) MZK_HIP(hipEventCreateWithFlags(&ce.ready, hipEventDisableTiming));
    const bool hit = ce.valid && ce.gen == ws_generation() && ce.logn == logn && ce.lgn0 == lgn0 && memcmp(ce.off, offset_host, 8 * hf->nl) == 0;
    if (!hit) {
      // a regrown slot or another size invalidates the pointers: wait for earlier users of the old contents
      prof_begin(s, MZK_PH_NTT_PRESCALE);
      const unsigned bc = (unsigned)((ncol + GEN_CHUNK * 64 - 1) / (GEN_CHUNK * 64)), br = (unsigned)((nrow + GEN_CHUNK * 64 - 1) / (GEN_CHUNK * 64));
      if (fid == MZK_FIELD_M128) {
        hipLaunchKernelGGL((k_gen_pow_table<M128Params>), dim3(bc), dim3(64), 0, s, offw, (uint64_t)1, ncol, pre_col);
        hipLaunchKernelGGL((k_gen_pow_table<M128Params>), dim3(br), dim3(64), 0, s, offw, (uint64_t)ncol, nrow, pre_row);
      } else {
        hipLaunchKernelGGL((k_gen_pow_table<FrParams>), dim3(bc), dim3(64), 0, s, offw, (uint64_t)1, ncol, pre_col);
        hipLaunchKernelGGL((k_gen_pow_table<FrParams>), dim3(br), dim3(64), 0, s, offw, (uint64_t)ncol, nrow, pre_row);
      }
      MZK_HIP(hipGetLastError());
      prof_end(s, MZK_PH_NTT_PRESCALE);
      memset(ce.off, 0, sizeof ce.off);
      memcpy(ce.off, offset_host, 8 * hf->nl);
      ce.logn = logn; ce.lgn0 = lgn0; ce.gen = ws_generation(); ce.valid = true;
      MZK_HIP(hipEventRecord(ce.ready, s));
    } else {
      MZK_HIP(hipStreamWaitEvent(s, ce.ready, 0));
    }
    const PreArgs pre{(const u32*)d_coef, n_coef, pre_row, pre_col};
    if (fid == MZK_FIELD_M128) return run_plan<M128Params>(pl, (const u32*)d_coef, (u32*)d_out, s, &pre, batch);
    return run_plan<FrParams>(pl, (const u32*)d_coef, (u32*)d_out, s, &pre, batch);
  }
  void* scaled = nullptr;
  MZK_TRY(ws_get(WS_NTT_IO_A, batch * order * field_bytes(fid), &scaled));
  const size_t chunks_per = (order + GEN_CHUNK - 1) / GEN_CHUNK;
  const unsigned blocks = (unsigned)((chunks_per * batch + 255) / 256);
  prof_begin(s, MZK_PH_NTT_PRESCALE);
  if (fid == MZK_FIELD_M128)
    hipLaunchKernelGGL((k_coset_scale_pad<M128Params>), dim3(blocks), dim3(256), 0, s, (const u32*)d_coef, n_coef, offw, (u32*)scaled, order, chunks_per, batch);
  else
    hipLaunchKernelGGL((k_coset_scale_pad<FrParams>), dim3(blocks), dim3(256), 0, s, (const u32*)d_coef, n_coef, offw, (u32*)scaled, order, chunks_per, batch);
  MZK_HIP(hipGetLastError());
  prof_end(s, MZK_PH_NTT_PRESCALE);
  return ntt_batch_dev_impl(fid, generator_host, scaled, d_out, order, batch, 0, s);
}

// FRI split-and-fold (zkstark/fri.rs:182-193).  With q_i = alpha / (offset omega^i):
//   out[i] = 2^-1 ((1 + q_i) a + (1 - q_i) b) = 2^-1 (a + b) + r_i (a - b),   r_i = 2^-1 alpha offset^-1 omega^-i.
// r_0 and omega^-1 are host parameters (Montgomery form); each lane walks per_lane consecutive i.
template <class P>
__global__ void k_fri_fold(const u32* __restrict__ cw, size_t h, Words8 r0_mont, Words8 winv_mont, Words8 half_mont,
                           u32* __restrict__ out, int per_lane) {
  const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i0 = chunk * (size_t)per_lane;
  if (i0 >= h) return;
  const Fe<P> winv = fe_unpack<P>(winv_mont.w), half = fe_unpack<P>(half_mont.w);
  Fe<P> r = FeAsm<P>::mul(fe_unpack<P>(r0_mont.w), fe_pow_u64<P>(winv, i0));   // r_{i0}, Montgomery form
  for (int t = 0; t < per_lane && i0 + t < h; t++) {
    const Fe<P> a = gload<P>(cw, i0 + t), b = gload<P>(cw, h + i0 + t);     // canonical, plain domain
    const Fe<P> sum = fe_add<P>(a, b);                                       // < 2p, limbs < 2^30
    const Fe<P> dif = fe_carry<P>(fe_sub<P, 4>(a, b));                       // a - b + 4p
    const Fe<P> o = fe_add<P>(FeAsm<P>::mul(sum, half), FeAsm<P>::mul(dif, r));      // plain * Montgomery constant = plain
    gstore<P>(out, i0 + t, fe_reduce<P>(o));                                 // .sanitize()
    r = FeAsm<P>::mul(r, winv);
  }
}
// The fold's host-side constants for (offset, omega): 2^-1, offset^-1, omega^-1 (plain) and R mod p.  FRI::commit squares offset and
// omega from round to round (fri.rs:186-187), and so it may their inverses: one set of inversions per commit instead of per round.
int fri_fold_consts(int fid, const uint64_t* offset, const uint64_t* omega, FriFoldConsts* fc) {
  const HostField* hf = host_field(fid);
  uint64_t two[4] = {2, 0, 0, 0};
  h_invmod(hf, fc->half, two);
  h_invmod(hf, fc->oinv, offset);
  h_invmod(hf, fc->winv, omega);
  h_powmod_u64(hf, fc->rmod, two, (uint64_t)(29 * (fid == MZK_FIELD_M128 ? 5 : 9)));
  return MZK_OK;
}
void fri_fold_consts_square(int fid, FriFoldConsts* fc) {
  const HostField* hf = host_field(fid);
  h_mulmod(hf, fc->oinv, fc->oinv, fc->oinv);
  h_mulmod(hf, fc->winv, fc->winv, fc->winv);
}
int fri_fold_dev_consts(int fid, const void* d_cw, size_t n, const uint64_t* alpha, const FriFoldConsts& fc, void* d_out, hipStream_t s) {
  const size_t h = n / 2;
  if (h == 0) return MZK_OK;
  const HostField* hf = host_field(fid);
  uint64_t r0[4], winv[4], halfv[4];
  h_mulmod(hf, r0, fc.half, alpha);
  h_mulmod(hf, r0, r0, fc.oinv);
  h_mulmod(hf, r0, r0, fc.rmod); h_mulmod(hf, winv, fc.winv, fc.rmod); h_mulmod(hf, halfv, fc.half, fc.rmod);
  Words8 r0w, winvw, halfw;
  to_words(r0, hf->nl, &r0w); to_words(winv, hf->nl, &winvw); to_words(halfv, hf->nl, &halfw);
  const int per_lane = h >= ((size_t)1 << 20) ? GEN_CHUNK : (h >= ((size_t)1 << 16) ? 4 : 1);
  const size_t chunks = (h + (size_t)per_lane - 1) / (size_t)per_lane;
  const unsigned blocks = (unsigned)((chunks + 127) / 128);
  if (fid == MZK_FIELD_M128)
    hipLaunchKernelGGL((k_fri_fold<M128Params>), dim3(blocks), dim3(128), 0, s, (const u32*)d_cw, h, r0w, winvw, halfw, (u32*)d_out, per_lane);
  else
    hipLaunchKernelGGL((k_fri_fold<FrParams>), dim3(blocks), dim3(128), 0, s, (const u32*)d_cw, h, r0w, winvw, halfw, (u32*)d_out, per_lane);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int fri_fold_dev_impl(int fid, const void* d_cw, size_t n, const uint64_t* alpha, const uint64_t* offset, const uint64_t* omega,
                      void* d_out, hipStream_t s) {
  if (fid != MZK_FIELD_FR && fid != MZK_FIELD_M128) { set_error("fri_fold: bad field id %d", fid); return MZK_E_ARG; }
  const size_t h = n / 2;
  if (h == 0) return MZK_OK;
  if (!d_cw || !d_out || !alpha || !offset || !omega) { set_error("fri_fold: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(fid);
  if (!h_is_canonical(hf, alpha) || !h_is_canonical(hf, offset) || !h_is_canonical(hf, omega)) { set_error("fri_fold: parameter not canonical"); return MZK_E_RANGE; }
  // host parameter math: 2^-1, offset^-1, omega^-1, all to Montgomery form (x * R mod p)
  uint64_t two[4] = {2, 0, 0, 0}, halfv[4], oinv[4], winv[4], r0[4], rmod[4];
  h_invmod(hf, halfv, two);
  h_invmod(hf, oinv, offset);
  h_invmod(hf, winv, omega);
  h_mulmod(hf, r0, halfv, alpha);
  h_mulmod(hf, r0, r0, oinv);
  h_powmod_u64(hf, rmod, two, (uint64_t)(29 * (fid == MZK_FIELD_M128 ? 5 : 9)));
  h_mulmod(hf, r0, r0, rmod); h_mulmod(hf, winv, winv, rmod); h_mulmod(hf, halfv, halfv, rmod);
  Words8 r0w, winvw, halfw;
  to_words(r0, hf->nl, &r0w); to_words(winv, hf->nl, &winvw); to_words(halfv, hf->nl, &halfw);
  // consecutive i per lane: a long walk amortises the lane's omega^-i0 power where there are lanes to spare; the late rounds
  // of a FRI commit are short codewords on an empty GPU, where the walk itself is the latency (16 steps: 18 us; one: 6 us)
  const int per_lane = h >= ((size_t)1 << 20) ? GEN_CHUNK : (h >= ((size_t)1 << 16) ? 4 : 1);
  const size_t chunks = (h + (size_t)per_lane - 1) / (size_t)per_lane;
  const unsigned blocks = (unsigned)((chunks + 127) / 128);
  if (fid == MZK_FIELD_M128)
    hipLaunchKernelGGL((k_fri_fold<M128Params>), dim3(blocks), dim3(128), 0, s, (const u32*)d_cw, h, r0w, winvw, halfw, (u32*)d_out, per_lane);
  else
    hipLaunchKernelGGL((k_fri_fold<FrParams>), dim3(blocks), dim3(128), 0, s, (const u32*)d_cw, h, r0w, winvw, halfw, (u32*)d_out, per_lane);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// out[i] = a[i] * b[i]^-1 with inverse(0) = 0 (field.rs:209-232: the extended Euclid returns t = 0 for 0), plain
// domain in and out.  One lane owns DIV_BATCH consecutive elements and inverts them together (Montgomery's
// trick: running products, ONE Fermat inversion, unwind): ~5 products per element + 1/DIV_BATCH of an inversion.
constexpr int DIV_BATCH = 16;
// regs > 1: `regs` numerator vectors (a_stride elements apart) over ONE denominator vector -- the interpolation of several trace
// registers over one domain divides every register's values by the same Z'(x_i) (ntt.rs:233-242 per register): the lane's inversion
// serves all of them, out[r * out_stride + i] = a[r * a_stride + i] / b[i].
template <class P>
__global__ __launch_bounds__(128) void k_pointwise_div(const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n,
                                                        size_t regs = 1, size_t a_stride = 0, size_t out_stride = 0) {
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * DIV_BATCH;
  if (i0 >= n) return;
  const int len = (int)((n - i0 < (size_t)DIV_BATCH) ? (n - i0) : (size_t)DIV_BATCH);
  Fe<P> pre[DIV_BATCH];                                      // lives in scratch (dynamically indexed): 36 B per element
  Fe<P> run = fe_one<P>();
  for (int j = 0; j < len; j++) {
    {
      const Fe<P> y = gload<P>(b, i0 + j);                   // canonical
      pre[j] = run;                                          // product of the earlier non-zero denominators (Montgomery)
      Fe<P> ym = fe_to_mont<P>(y);
      if (fe_is_zero_canon<P>(y)) ym = fe_one<P>();
      run = FeAsm<P>::mul(run, ym);
    }
  }
  Fe<P> inv = fe_inv<P>(run);
  for (int j = len - 1; j >= 0; j--) {
    {
      const Fe<P> y = gload<P>(b, i0 + j);
      const bool zero = fe_is_zero_canon<P>(y);
      const Fe<P> yinv = FeAsm<P>::mul(inv, pre[j]);             // Montgomery form of 1 / y_j
      Fe<P> ym = fe_to_mont<P>(y);
      if (zero) ym = fe_one<P>();
      inv = FeAsm<P>::mul(inv, ym);
      for (size_t r = 0; r < regs; r++) {
        const Fe<P> x = gload<P>(a, r * a_stride + i0 + j);
        const Fe<P> q = fe_reduce<P>(FeAsm<P>::mul(x, yinv));       // plain * Montgomery = plain
        gstore<P>(out, r * out_stride + i0 + j, zero ? fe_zero<P>() : q);
      }
    }
  }
}

// ntt::fast_coset_divide's transform part (ntt.rs:304-329): lhs (tl coefficients), rhs (tr) scaled by offset^i and
// padded to `order`, forward transforms, pointwise quotient, inverse transform, the first tl - tr + 1 coefficients
// scaled by offset^-i.  root has order exactly `order` (the caller squared it down, ntt.rs:299-302).
int coset_divide_dev_impl(int fid, const void* d_lhs, size_t tl, const void* d_rhs, size_t tr, const uint64_t* offset_host,
                          const uint64_t* root_host, size_t order, void* d_out, hipStream_t s) {
  const HostField* hf = host_field(fid);
  const size_t esz = field_bytes(fid);
  void *ea, *eb;
  MZK_TRY(ws_get(WS_MISC_A, order * esz, &ea));
  MZK_TRY(ws_get(WS_MISC_B, order * esz, &eb));
  MZK_TRY(coset_lde_dev_impl(fid, d_lhs, tl, offset_host, root_host, ea, order, s));
  MZK_TRY(coset_lde_dev_impl(fid, d_rhs, tr, offset_host, root_host, eb, order, s));
  const size_t lanes = (order + DIV_BATCH - 1) / DIV_BATCH;
  const unsigned blocks = (unsigned)((lanes + 127) / 128);
  if (fid == MZK_FIELD_M128)
    hipLaunchKernelGGL((k_pointwise_div<M128Params>), dim3(blocks), dim3(128), 0, s, (const u32*)ea, (const u32*)eb, (u32*)ea, order);
  else
    hipLaunchKernelGGL((k_pointwise_div<FrParams>), dim3(blocks), dim3(128), 0, s, (const u32*)ea, (const u32*)eb, (u32*)ea, order);
  MZK_HIP(hipGetLastError());
  MZK_TRY(ntt_dev_impl(fid, root_host, ea, eb, order, 1, nullptr, s));
  uint64_t oinv[4];
  h_invmod(hf, oinv, offset_host);
  Words8 ow;
  to_words(oinv, hf->nl, &ow);
  const size_t ql = tl - tr + 1;
  const size_t chunks = (ql + GEN_CHUNK - 1) / GEN_CHUNK;
  const unsigned sblocks = (unsigned)((chunks + 255) / 256);
  if (fid == MZK_FIELD_M128)
    hipLaunchKernelGGL((k_coset_scale_pad<M128Params>), dim3(sblocks), dim3(256), 0, s, (const u32*)eb, ql, ow, (u32*)d_out, ql, chunks, (size_t)1);
  else
    hipLaunchKernelGGL((k_coset_scale_pad<FrParams>), dim3(sblocks), dim3(256), 0, s, (const u32*)eb, ql, ow, (u32*)d_out, ql, chunks, (size_t)1);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// out[i] = a[i] / b[i] with inverse(0) = 0 (field.rs:209-232)
int pointwise_div_dev(int fid, const void* d_a, const void* d_b, void* d_out, size_t n, hipStream_t s) {
  if (n == 0) return MZK_OK;
  const unsigned blocks = (unsigned)(((n + DIV_BATCH - 1) / DIV_BATCH + 127) / 128);
  if (fid == MZK_FIELD_M128) hipLaunchKernelGGL((k_pointwise_div<M128Params>), dim3(blocks), dim3(128), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n);
  else hipLaunchKernelGGL((k_pointwise_div<FrParams>), dim3(blocks), dim3(128), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
// out[r * out_stride + i] = a[r * a_stride + i] * b[i]: `regs` vectors times ONE vector (the interpolation plan keeps 1 / Z'(d_i), so a
// call multiplies where it used to divide: the shared inversion chain was 0.18 ms of a 1.5-ms interpolation)
template <class P>
__global__ __launch_bounds__(256) void k_pointwise_mul_shared(const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n, size_t regs,
                                                              size_t a_stride, size_t out_stride) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe<P> ym = fe_to_mont<P>(gload<P>(b, i));
  for (size_t r = 0; r < regs; r++) gstore<P>(out, r * out_stride + i, fe_reduce<P>(FeAsm<P>::mul(gload<P>(a, r * a_stride + i), ym)));      // plain * Montgomery = plain
}
int pointwise_mul_shared_dev(int fid, const void* d_a, size_t a_stride, const void* d_b, void* d_out, size_t out_stride, size_t n, size_t regs, hipStream_t s) {
  if (n == 0 || regs == 0) return MZK_OK;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (fid == MZK_FIELD_M128) hipLaunchKernelGGL((k_pointwise_mul_shared<M128Params>), dim3(blocks), dim3(256), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n, regs, a_stride, out_stride);
  else hipLaunchKernelGGL((k_pointwise_mul_shared<FrParams>), dim3(blocks), dim3(256), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n, regs, a_stride, out_stride);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
// `regs` numerator vectors (a_stride elements apart) over one denominator vector; results out_stride elements apart
int pointwise_div_shared_dev(int fid, const void* d_a, size_t a_stride, const void* d_b, void* d_out, size_t out_stride, size_t n, size_t regs, hipStream_t s) {
  if (n == 0 || regs == 0) return MZK_OK;
  const unsigned blocks = (unsigned)(((n + DIV_BATCH - 1) / DIV_BATCH + 127) / 128);
  if (fid == MZK_FIELD_M128) hipLaunchKernelGGL((k_pointwise_div<M128Params>), dim3(blocks), dim3(128), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n, regs, a_stride, out_stride);
  else hipLaunchKernelGGL((k_pointwise_div<FrParams>), dim3(blocks), dim3(128), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n, regs, a_stride, out_stride);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int pointwise_mul_dev(int fid, const void* d_a, const void* d_b, void* d_out, size_t n, hipStream_t s) {
  if (n == 0) return MZK_OK;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (fid == MZK_FIELD_M128)
    hipLaunchKernelGGL((k_pointwise_mul<M128Params>), dim3(blocks), dim3(256), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n);
  else
    hipLaunchKernelGGL((k_pointwise_mul<FrParams>), dim3(blocks), dim3(256), 0, s, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, n);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

}  // namespace mzk
