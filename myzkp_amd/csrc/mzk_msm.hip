// mzk_msm.hip -- Pippenger bucket multi-scalar multiplication over BN254 G1 on gfx950.
//
// Stands behind Polynomial::eval_with_powers_on_curve (myzkp/src/modules/algebra/polynomial.rs:156-165)
// = commit_kzg (algebra/kzg.rs:57-59) and the second MSM of open_kzg (kzg.rs:70).  The reference runs
// n independent affine double-and-add scalar multiplications (curve/curve.rs:168-191); this computes
// the same group element sum_i s_i P_i with the bucket method and returns the same canonical affine
// point (or the all-zero infinity encoding).
//
// Pipeline, all on the device:
//   0. k_prepare_points    affine canonical -> Montgomery words (generic path; an SRS handle holds
//                          Montgomery-form window tables T[w][i] = 2^(16 w) P_i instead: k_srs_tables)
//   1. k_digits_count[_lds] scalars -> signed c-bit digits (sanitize = reduce mod r first,
//                          polynomial.rs:162), histogram of buckets; the atomic's return value is the
//                          entry's rank (LDS atomics when the whole bucket set fits the 160 KiB LDS)
//   2. scan                exclusive prefix sum of the histogram
//   3. k_digits_scatter[_lds] counting-sort placement of (point reference, sign), no atomics
//   4. k_seg_accumulate    one lane per fixed-size segment of the sorted entries, XYZZ += affine
//                          (madd-2008-s), exception-complete; k_seg_combine sums a bucket's partials
//   5. k_halve_step / k_reduce_tail   sum_b (b+1) B_b per bucket set by in-place halving
//   6. k_window_combine (mzk_msm_tail.hip)  Horner over the bucket sets, XYZZ -> affine (one inversion)
#include <stdlib.h>
#include <type_traits>
#include "mzk_common.h"
#include "mzk_ec.h"
#include "mzk_coop.h"
#include "mzk_row.h"
#include "mzk_glv.h"
#include "mzk_affine_wave.h"

namespace mzk {

constexpr int SCALAR_BITS = 254;
constexpr int MAX_WINDOWS = 32;

struct MsmShape {
  int c;          // window bits
  int nwin;       // windows
  int lgB;        // log2 buckets per window = c - 1
  size_t nbuckets;
};

static MsmShape choose_shape(size_t n) {
  int lg = 0;
  while (((size_t)1 << lg) < n) lg++;
  int c = lg - 3;
  if (c < 8) c = 8;
  if (c > 16) c = 16;
  MsmShape s;
  s.c = c;
  s.nwin = SCALAR_BITS / c + 1;
  s.lgB = c - 1;
  s.nbuckets = (size_t)s.nwin << s.lgB;
  return s;
}

// Generic layout after the GLV split: 2n points with scalars below 2^126.  Windows must cover 127 bits plus the
// signed-digit carry; at c = 16 that is exactly 8 windows (2^18 buckets: the two-level sort's power of two).
constexpr int GLV_MAG_BITS = 126;      // |k1|, |k2| < 2^126 (mzk_glv.h)
static MsmShape choose_shape_glv(size_t n) {
  int lg = 0;
  while (((size_t)1 << lg) < 2 * n) lg++;
  int c = lg - 3;
  if (c < 8) c = 8;
  if (c > 16) c = 16;
  if (c == 15) c = 16;     // 2^17 pairs: 8 full windows + the two-level sort beat 9 windows with a 6-bit top window
  // From 3 x 2^21 pairs on: 19 bits -- SEVEN windows per half instead of eight (14 mixed additions per pair, not 16: the accumulate is
  // 80 % of the call), 7 x 2^18 buckets.  17 and 18 bits still need eight windows (7 x 18 = 126 leaves the top window nothing but
  // carries), 20 bits also seven but twice the buckets, 22 bits six windows over 6 x 2^21 buckets whose reduction and one-pass sort cost
  // more than the windows save.  The larger bucket set costs ~0.7 ms more to sort, combine and reduce whatever n is, the saved additions
  // 0.145 ms per 2^20 pairs: 2^22 +3.6 %, 2^23 -2 %, 2^24 -6.5 %, 2^26 -7.5 % (profiles/round6_generic_window_sweep.txt).
  static const int env_glv_c = tune_int("MZK_GLV_C", 0);         // tuning build: force a width (the sweep)
  if (n >= ((size_t)3 << 21)) c = 19;
  if (env_glv_c > 0) c = env_glv_c;
  // nwin windows must cover the 126 magnitude bits plus the signed-digit carry.  The top window only holds
  // 126 - c (nwin - 1) real bits; if that is (almost) nothing, every scalar whose carry runs into it lands in the
  // same few buckets (c = 14: ONE bucket receives a third of all entries) -- step c down until the top window is
  // reasonably populated.
  for (; c > 8; c--) {
    const int nw = GLV_MAG_BITS / c + 1;
    if (GLV_MAG_BITS - c * (nw - 1) >= 5) break;
  }
  MsmShape s;
  s.c = c;
  s.nwin = GLV_MAG_BITS / c + 1;
  s.lgB = c - 1;
  s.nbuckets = (size_t)s.nwin << s.lgB;
  return s;
}

int msm_generic_window_bits(size_t n) { return choose_shape_glv(n ? n : 1).c; }

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

// Segment partials travel from k_seg_accumulate to k_seg_combine as RAW limbs (4 x 9 words = 144 bytes, coordinate-major):
// a flush then costs 9 stores instead of four canonical reductions + packing (~700 instructions, a quarter of a mixed
// addition).  That matters because a wave pays for a flush whenever ANY of its lanes crosses a bucket boundary -- nearly
// every iteration in the generic layout (64-entry buckets, 64-entry segments).  Limbs are whatever the accumulator held:
// normalised, value < 2.5 p, exactly what xyzz_add accepts; infinity is all-zero.
constexpr int SLOT_WORDS = 4 * FqParams::L;      // 36
__device__ __forceinline__ void xyzz_gstore_raw(u32* __restrict__ g, size_t idx, const Xyzz& p) {
  u32 w[SLOT_WORDS];
  const bool inf = xyzz_is_inf(p);
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) {
    w[i] = inf ? 0u : p.X.l[i]; w[9 + i] = inf ? 0u : p.Y.l[i]; w[18 + i] = inf ? 0u : p.ZZ.l[i]; w[27 + i] = inf ? 0u : p.ZZZ.l[i];
  }
  uint4* p4 = reinterpret_cast<uint4*>(g + idx * SLOT_WORDS);
#pragma unroll
  for (int q = 0; q < SLOT_WORDS / 4; q++) p4[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
__device__ __forceinline__ Xyzz xyzz_gload_raw(const u32* __restrict__ g, size_t idx) {
  u32 w[SLOT_WORDS];
  const uint4* p4 = reinterpret_cast<const uint4*>(g + idx * SLOT_WORDS);
#pragma unroll
  for (int q = 0; q < SLOT_WORDS / 4; q++) { const uint4 v = p4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
  Xyzz p;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) { p.X.l[i] = w[i]; p.Y.l[i] = w[9 + i]; p.ZZ.l[i] = w[18 + i]; p.ZZZ.l[i] = w[27 + i]; }
  return p;
}
// quad form: lane k fetches coordinate k (9 words) and the quad exchanges them
__device__ __forceinline__ Xyzz xyzz_gload_raw_quad(const u32* __restrict__ g, size_t idx, int lane) {
  Fq mine;
  const u32* src = g + idx * SLOT_WORDS + 9 * lane;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) mine.l[i] = src[i];
  Xyzz p;
  p.X = quad_bcast<0>(mine);
  p.Y = quad_bcast<1>(mine);
  p.ZZ = quad_bcast<2>(mine);
  p.ZZZ = quad_bcast<3>(mine);
  return p;
}

// ---- 0. point preparation ------------------------------------------------------------------------------
__global__ void k_prepare_points(const u32* __restrict__ in, u32* __restrict__ out, size_t n, u32* __restrict__ phi_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[16];
  load_words8(in + i * 16, w);
  load_words8(in + i * 16 + 8, w + 8);
  Affine a;
  const bool inf = affine_words_is_inf(w);
  if (!inf) {
    a = affine_load_plain(w);
    affine_store_mont(a, w);
  }
  store_words8(out + i * 16, w);
  store_words8(out + i * 16 + 8, w + 8);
  if (phi_out) {        // phi(x, y) = (beta x, y); infinity stays the all-zero record
    if (!inf) {
      Fq beta;
#pragma unroll
      for (int k = 0; k < FqParams::L; k++) beta.l[k] = GlvParams::BETA_MONT[k];
      a.x = fe_reduce<FqParams>(fe_mul<FqParams>(a.x, beta));
      affine_store_mont(a, w);
    }
    store_words8(phi_out + i * 16, w);
    store_words8(phi_out + i * 16 + 8, w + 8);
  }
}
// phi images of already prepared (Montgomery) points
__global__ void k_phi_points(const u32* __restrict__ in_mont, size_t n, u32* __restrict__ phi_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[16];
  load_words8(in_mont + i * 16, w);
  load_words8(in_mont + i * 16 + 8, w + 8);
  if (!affine_words_is_inf(w)) {
    Affine a = affine_load_mont(w);
    Fq beta;
#pragma unroll
    for (int k = 0; k < FqParams::L; k++) beta.l[k] = GlvParams::BETA_MONT[k];
    a.x = fe_reduce<FqParams>(fe_mul<FqParams>(a.x, beta));
    affine_store_mont(a, w);
  }
  store_words8(phi_out + i * 16, w);
  store_words8(phi_out + i * 16 + 8, w + 8);
}

// Montgomery (prepared) points back to the ABI form (SRS dump / download)
__global__ void k_points_to_plain(const u32* __restrict__ in_mont, size_t n, u32* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[16];
  load_words8(in_mont + i * 16, w);
  load_words8(in_mont + i * 16 + 8, w + 8);
  if (!affine_words_is_inf(w)) affine_store_plain(affine_load_mont(w), w);
  store_words8(out + i * 16, w);
  store_words8(out + i * 16 + 8, w + 8);
}

// ---- 1/3. signed-digit decomposition ----------------------------------------------------------------
// Canonical scalar words (sanitize(): polynomial.rs:162 -> field.rs:260-270).
__device__ __forceinline__ void load_scalar_canonical(const u32* __restrict__ scalars, size_t i, u32* w) {
  load_words8(scalars + i * 8, w);
  // fast path: top word below r's top word => already canonical
  if (w[7] >= 0x30644e72u) {
    Fe<FrParams> x = fe_reduce<FrParams>(fe_unpack<FrParams>(w));
    fe_pack<FrParams>(x, w);
  }
}
// digit of window `win` before carry handling: bits [c win, c win + c)
__device__ __forceinline__ u32 raw_window(const u32* w, int win, int c) {
  const int bit = win * c;
  if (bit >= 256) return 0;
  const int k = bit >> 5, s = bit & 31;
  u64 v = w[k];
  if (k + 1 < 8) v |= (u64)w[k + 1] << 32;
  return (u32)(v >> s) & ((1u << c) - 1u);
}
// Bucket layout.  Generic MSM: bucket = window * 2^(c-1) + |digit| - 1, entry = point index.
// Fixed-base MSM over precomputed tables T[w][i] = 2^(c w) P_i: every window shares ONE bucket set,
// bucket = |digit| - 1, entry = w * table_stride + i.
struct DigitLayout {
  int c, nwin, merged;
  int sets;              // merged layout: bucket sets (mzk_srs::sets); window w goes to set w % sets and reads table row w / sets
  size_t table_stride;
  int glv;               // generic layout only: scalars are GLV-split, the phi images of the points start at phi_offset
  size_t phi_offset;
};
// Calls emit(slot, key, payload) for every non-zero signed digit of scalar i (canonical words w).  `slot` numbers
// the (half, window) positions of a scalar: 0 .. slots_per_scalar-1.  key = bucket index; payload = point reference
// with the sign in bit 31.
//   merged : windows of the full scalar, key = |d| - 1, reference = window * table_stride + i
//   generic: k = k1 + k2 lambda (mzk_glv.h); windows of |k1| address point i, windows of |k2| its phi image
//            phi_offset + i; key = window * 2^(c-1) + |d| - 1; the sign of the part flips the digit's sign
__device__ __forceinline__ int slots_per_scalar(const DigitLayout& L) { return L.glv ? 2 * L.nwin : L.nwin; }
template <class Emit>
__device__ __forceinline__ void walk_digits(const u32* w, const DigitLayout& L, size_t i, Emit emit) {
  const int c = L.c;
  const u32 half = 1u << (c - 1);
  if (!L.glv) {
    u32 carry = 0;
    for (int win = 0; win < L.nwin; win++) {
      u32 raw = raw_window(w, win, c) + carry;
      u32 neg = 0, mag = raw;
      carry = 0;
      if (raw > half) { mag = (1u << c) - raw; neg = 1; carry = 1; }
      if (mag != 0) {
        const u32 key = (L.merged ? ((u32)(win % L.sets) << (c - 1)) : ((u32)win << (c - 1))) + (mag - 1);
        const u32 payload = (L.merged ? (u32)((size_t)(win / L.sets) * L.table_stride + i) : (u32)i) | (neg << 31);
        emit(win, key, payload);
      }
    }
    return;
  }
  u32 m[2][8], sg[2];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int j = 4; j < 8; j++) m[h][j] = 0;
  glv_split(w, m[0], &sg[0], m[1], &sg[1]);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    u32 carry = 0;
    for (int win = 0; win < L.nwin; win++) {
      u32 raw = raw_window(m[h], win, c) + carry;
      u32 neg = 0, mag = raw;
      carry = 0;
      if (raw > half) { mag = (1u << c) - raw; neg = 1; carry = 1; }
      if (mag != 0) {
        const u32 key = ((u32)win << (c - 1)) + (mag - 1);
        const u32 payload = (u32)(i + (h ? L.phi_offset : 0)) | ((neg ^ sg[h]) << 31);
        emit(h * L.nwin + win, key, payload);
      }
    }
  }
}

// Signed c-bit digits without the serial carry walk: with t = k + sum_w (2^(c-1) - 1) 2^(c w) the digit of window w is
// window_w(t) - (2^(c-1) - 1) (the carries of that ONE long addition are exactly the recoding's carries: window w overflows iff
// raw_w + carry > 2^(c-1)), same digits as walk_digits.  Word k of the constant, C a compile-time width:
constexpr u32 digit_bias_word(int C, int k) {
  const int nwin = 254 / C + 1;
  const unsigned long long hm1 = (1ull << (C - 1)) - 1;
  unsigned long long acc = 0;
  for (int w = 0; w < nwin; w++) {
    const int sh = w * C - 32 * k;
    if (sh >= 0 && sh < 32) acc |= (hm1 << sh) & 0xffffffffull;
    else if (sh < 0 && sh > -32) acc |= hm1 >> (-sh);
  }
  return (u32)acc;
}
// walk_digits for the merged layout with a compile-time window width: the same (window, key, payload) triples in the same order,
// from ONE long addition and NWIN independent extractions with static word indices (walk_digits' runtime window index makes
// every word access a select chain).  Used by the coarse passes of the two-level sort and by the sortless small path.
template <int C, class Emit>
__device__ __forceinline__ void walk_digits_merged(const u32* w, size_t table_stride, size_t i, Emit emit) {
  constexpr int NWIN = 254 / C + 1;
  constexpr u32 HALF = 1u << (C - 1), MASKC = (1u << C) - 1u;
  u32 t[9];
  u64 cy = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { cy += (u64)w[k] + digit_bias_word(C, k); t[k] = (u32)cy; cy >>= 32; }
  t[8] = (u32)cy + digit_bias_word(C, 8);
#pragma unroll
  for (int win = 0; win < NWIN; win++) {
    const int bit = win * C, k = bit >> 5, sft = bit & 31;
    const u64 pair = (u64)t[k] | ((k + 1 < 9) ? ((u64)t[k + 1] << 32) : 0ull);
    const u32 v = (u32)(pair >> sft) & MASKC;            // digit + HALF - 1
    if (v == HALF - 1u) continue;                        // digit 0
    const bool neg = v < HALF - 1u;
    const u32 mag = neg ? (HALF - 1u) - v : v - (HALF - 1u);
    emit(win, mag - 1u, (u32)((size_t)win * table_stride + i) | ((u32)neg << 31));
  }
}

// Exclusive prefix of one value per lane over a workgroup of NT lanes, and the workgroup's total: shuffles inside the waves, the
// NT / 64 wave totals by the first wave -- two barriers (the Hillis-Steele form over LDS this replaces took 2 log2(NT) = 20 at 1024
// lanes, ~5 us of a kernel that runs for 20-50).  sc: NT / 64 + 1 words of LDS; a barrier must separate two calls on the same sc.
template <int NT>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* sc, u32* total) {
  constexpr int NW = NT / 64;
  static_assert(NT % 64 == 0 && NW <= 64, "whole waves, at most 64 of them");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const u32 t = (u32)__shfl_up((int)incl, off, 64); if (lane >= off) incl += t; }
  if (lane == 63) sc[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    u32 t = (lane < NW) ? sc[lane] : 0u;
#pragma unroll
    for (int off = 1; off < NW; off <<= 1) { const u32 u = (u32)__shfl_up((int)t, off, 64); if (lane >= off) t += u; }
    if (lane < NW) sc[lane] = t;
  }
  __syncthreads();
  if (total) *total = sc[NW - 1];
  return (wave ? sc[wave - 1] : 0u) + incl - v;
}

// Counter increment (LDS or global) that stays fast when most of a wave hits ONE counter (bit-vector or repeated
// scalars, the sparsely populated top window of a GLV half):
// the lanes sharing the first active lane's key take a single atomic together.  Returns the lane's rank.
__device__ __forceinline__ u32 counter_inc_agg(u32* __restrict__ ctr, u32 key) {
  const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)key);
  const u64 same = __builtin_amdgcn_ballot_w64(key == k0);
  if (__builtin_popcountll(same) >= 16) {
    u32 r;
    if (key == k0) {
      const u32 lane_id = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const u32 rank = (u32)__builtin_popcountll(same & (((u64)1 << lane_id) - 1));
      u32 base = 0;
      if (rank == 0) base = atomicAdd(&ctr[k0], (u32)__builtin_popcountll(same));
      base = (u32)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(same));
      r = base + rank;
    } else {
      r = atomicAdd(&ctr[key], 1u);
    }
    return r;
  }
  return atomicAdd(&ctr[key], 1u);
}
constexpr u32 NO_RANK = 0xffffffffu;

// Pass 1: histogram; the value returned by the atomic is this entry's rank inside its bucket, kept
// (coalesced, [slot][i]) so that the scatter pass needs no atomics.
__global__ __launch_bounds__(256) void k_digits_count(const u32* __restrict__ scalars, size_t n, DigitLayout L, u32* __restrict__ counts,
                                                       u32* __restrict__ ranks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  load_scalar_canonical(scalars, i, w);
  walk_digits(w, L, i, [&](int slot, u32 key, u32) { ranks[(size_t)slot * n + i] = counter_inc_agg(counts, key); });
}
// Pass 2: entries[offsets[bucket] + rank] = reference | sign << 31
__global__ __launch_bounds__(256) void k_digits_scatter(const u32* __restrict__ scalars, size_t n, DigitLayout L, const u32* __restrict__ offsets,
                                                         const u32* __restrict__ ranks, u32* __restrict__ entries) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  load_scalar_canonical(scalars, i, w);
  walk_digits(w, L, i, [&](int slot, u32 key, u32 payload) { entries[offsets[key] + ranks[(size_t)slot * n + i]] = payload; });
}

// Merged (fixed-base) layout: one bucket set of 2^(c-1) = 32768 counters = 128 KiB fits the 160 KiB LDS of
// a CU, so the histogram runs on LDS atomics (the returned value is the entry's rank inside this
// workgroup's share of the bucket) and no global atomic is issued at all.  Workgroup g owns scalars
// [g * per_wg, (g+1) * per_wg); wg_hist[g][b] receives its counts.
constexpr int LDS_SORT_THREADS = 1024;
__global__ __launch_bounds__(LDS_SORT_THREADS) void k_digits_count_lds(const u32* __restrict__ scalars, size_t n, size_t per_wg, DigitLayout L,
                                                                      u32* __restrict__ wg_hist, u32* __restrict__ ranks) {
  extern __shared__ u32 hist[];
  const int NBK = 1 << (L.c - 1);
  for (int b = threadIdx.x; b < NBK; b += LDS_SORT_THREADS) hist[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const size_t hi = (lo + per_wg < n) ? lo + per_wg : n;
  const int c = L.c;
  const u32 half = 1u << (c - 1);
  for (size_t i = lo + threadIdx.x; i < hi; i += LDS_SORT_THREADS) {
    u32 w[8];
    load_scalar_canonical(scalars, i, w);
    u32 carry = 0;
    for (int win = 0; win < L.nwin; win++) {
      u32 raw = raw_window(w, win, c) + carry;
      u32 mag = raw;
      carry = 0;
      if (raw > half) { mag = (1u << c) - raw; carry = 1; }
      u32 rank = NO_RANK;
      if (mag != 0) rank = atomicAdd(&hist[mag - 1], 1u);
      ranks[(size_t)win * n + i] = rank;
    }
  }
  __syncthreads();
  u32* row = wg_hist + (size_t)blockIdx.x * NBK;
  for (int b = threadIdx.x; b < NBK; b += LDS_SORT_THREADS) row[b] = hist[b];
}
// counts[b] = sum_g wg_hist[g][b]; wg_hist[g][b] <- exclusive prefix over g (this workgroup's base inside
// bucket b)
__global__ __launch_bounds__(256) void k_wg_hist_prefix(u32* __restrict__ wg_hist, int nwg, int nbk, u32* __restrict__ counts) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbk) return;
  u32 run = 0;
  for (int g = 0; g < nwg; g++) {
    const u32 v = wg_hist[(size_t)g * nbk + b];
    wg_hist[(size_t)g * nbk + b] = run;
    run += v;
  }
  counts[b] = run;
}
__global__ __launch_bounds__(LDS_SORT_THREADS) void k_digits_scatter_lds(const u32* __restrict__ scalars, size_t n, size_t per_wg, DigitLayout L,
                                                                        const u32* __restrict__ offsets, const u32* __restrict__ wg_hist,
                                                                        const u32* __restrict__ ranks, u32* __restrict__ entries) {
  extern __shared__ u32 base[];   // offsets[b] + this workgroup's prefix inside bucket b
  const int NBK = 1 << (L.c - 1);
  const u32* row = wg_hist + (size_t)blockIdx.x * NBK;
  for (int b = threadIdx.x; b < NBK; b += LDS_SORT_THREADS) base[b] = offsets[b] + row[b];
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const size_t hi = (lo + per_wg < n) ? lo + per_wg : n;
  const int c = L.c;
  const u32 half = 1u << (c - 1);
  for (size_t i = lo + threadIdx.x; i < hi; i += LDS_SORT_THREADS) {
    u32 w[8];
    load_scalar_canonical(scalars, i, w);
    u32 carry = 0;
    for (int win = 0; win < L.nwin; win++) {
      u32 raw = raw_window(w, win, c) + carry;
      u32 neg = 0, mag = raw;
      carry = 0;
      if (raw > half) { mag = (1u << c) - raw; neg = 1; carry = 1; }
      if (mag != 0) {
        const u32 rank = ranks[(size_t)win * n + i];
        const u32 ref = (u32)((size_t)win * L.table_stride + i);
        entries[base[mag - 1] + rank] = ref | (neg << 31);
      }
    }
  }
}

// ---- 1b. two-level counting sort (large inputs) -----------------------------------------------------------
// A one-pass counting sort scatters 4-byte entries over the whole entry array (64 MiB at 2^20 pairs): every
// store dirties its own 32-byte sector, an 8x write amplification that bounded the single-pass kernels.  Two
// levels keep every store stream local:
//   coarse: 256 bins by the top 8 bits of the bucket key.  A workgroup owns 4096 scalars, counts its entries per
//           bin in LDS (k_coarse_count), a scan over [bin][workgroup] gives each (bin, workgroup) a contiguous
//           run, and k_coarse_scatter appends (payload, fine key) records to its runs through LDS cursors.
//   fine:   bin b is now a contiguous slice of ~E/256 records covering NB/256 buckets.  (bin, sub) workgroups
//           histogram their share in LDS (k_fine_count), a scan over [bucket][sub] yields the bucket offsets,
//           and k_fine_scatter writes the final 4-byte entries -- all inside one bin's region (256 KiB at 2^20),
//           which stays in L2 while it fills.
// key = bucket index: |digit| - 1 (merged layout) or window * 2^(c-1) + |digit| - 1.
#ifndef MZK_COARSE_LOG
#define MZK_COARSE_LOG 8
#endif
constexpr int COARSE_LOG = MZK_COARSE_LOG;      // 9: A/B build (one more bit for the point reference in the 4-byte sort records)
constexpr int COARSE_BINS = 1 << COARSE_LOG;
constexpr int COARSE_PER_WG = 4096;
constexpr int SORT2_THREADS = 1024;
// C: compile-time window width of the merged layout (walk_digits_merged), 0 = any layout by walk_digits
// CL: log2 of the coarse bins.  256 everywhere but at 20-bit windows (2^19 buckets, the default from 2^22 points on): there 1024, so
// that a bin covers 512 buckets instead of 2048 and a fine workgroup's round of 8192 records leaves in runs of ~16 entries (64 bytes)
// instead of ~4 -- k_fine_scatter cost 7 ps per entry at 2048 buckets per bin against 3.2 at 17-bit windows (profiles/round5_sort_1024_bins.txt).
// Scan-free form (bin_tot != null; profiles/round6_sort_without_global_scans_ab.txt): a bin's total is summed with one atomic per
// (workgroup, bin) into bin_tot (zeroed by the call's one memset), k_coarse_scatter* turns the totals into bin starts itself and
// CLAIMS its runs from per-bin cursors, and the workgroups also zero the fine level's per-bucket counters here (zero_words words at
// zero_ptr) -- the two launches of the global scan between this kernel and the scatter (~20 us whatever they scan) are gone.  Which
// workgroup's records come first inside a bin is then arbitrary, as the order of entries inside a bucket always was.
template <int C, int CL = COARSE_LOG>
__global__ __launch_bounds__(SORT2_THREADS) void k_coarse_count(const u32* __restrict__ scalars, size_t n, DigitLayout L, int key_shift,
                                                                 u32* __restrict__ binhist, int nwg, u32* __restrict__ bin_tot, u32* __restrict__ zero_ptr,
                                                                 size_t zero_words) {
  constexpr int BINS = 1 << CL;
  static_assert(BINS <= SORT2_THREADS, "one lane per bin");
  __shared__ u32 hist[BINS];
  if (threadIdx.x < BINS) hist[threadIdx.x] = 0;
  if (zero_ptr) {
    const size_t per = (zero_words + gridDim.x - 1) / gridDim.x;
    const size_t z0 = (size_t)blockIdx.x * per, z1 = (z0 + per < zero_words) ? z0 + per : zero_words;
    for (size_t z = z0 + threadIdx.x; z < z1; z += SORT2_THREADS) zero_ptr[z] = 0;
  }
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * COARSE_PER_WG;
  const size_t hi = (lo + COARSE_PER_WG < n) ? lo + COARSE_PER_WG : n;
  u32 w[COARSE_PER_WG / SORT2_THREADS][8];            // all of this lane's scalars in flight at once
#pragma unroll
  for (int k = 0; k < COARSE_PER_WG / SORT2_THREADS; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * SORT2_THREADS;
    if (i < hi) load_scalar_canonical(scalars, i, w[k]);
  }
#pragma unroll
  for (int k = 0; k < COARSE_PER_WG / SORT2_THREADS; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * SORT2_THREADS;
    if (i < hi) {
      auto emit = [&](int, u32 key, u32) { counter_inc_agg(hist, key >> key_shift); };
      if constexpr (C != 0) walk_digits_merged<C>(w[k], L.table_stride, i, emit);
      else walk_digits(w[k], L, i, emit);
    }
  }
  __syncthreads();
  if (threadIdx.x < BINS) {
    const u32 v = hist[threadIdx.x];
    binhist[(size_t)threadIdx.x * nwg + blockIdx.x] = v;
    if (bin_tot && v) atomicAdd(&bin_tot[threadIdx.x], v);
  }
}
// Exclusive prefix of the BINS bin totals into LDS (start[BINS] = all records), by the first wave: BINS / 64 consecutive bins per lane.
// Workgroup 0 also publishes it (start_out, BINS + 1 words): the fine kernels read the bin boundaries there.  Ends with a barrier.
template <int BINS>
__device__ __forceinline__ void bin_starts(const u32* __restrict__ bin_tot, u32* start, u32* __restrict__ start_out) {
  constexpr int PER = BINS / 64;
  static_assert(BINS % 64 == 0, "whole lanes");
  const int tid = threadIdx.x;
  if (tid < 64) {
    u32 c[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { c[q] = bin_tot[PER * tid + q]; sum += c[q]; }
    u32 incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u32 up = (u32)__shfl_up((int)incl, d, 64);
      if (tid >= d) incl += up;
    }
    u32 run = incl - sum;
#pragma unroll
    for (int q = 0; q < PER; q++) { start[PER * tid + q] = run; run += c[q]; }
    if (tid == 63) start[BINS] = incl;
  }
  __syncthreads();
  if (start_out && blockIdx.x == 0) for (int b = tid; b <= BINS; b += blockDim.x) start_out[b] = start[b];
}
// Intermediate records of the two-level sort: (payload, fine key).  The 8-byte form always works; when the point
// reference, the fine key and the sign fit 32 bits together (up to 2^20 pairs in both layouts) the 4-byte form halves the
// traffic of the coarse scatter and of both fine passes.
struct Rec8 {
  typedef uint2 T;
  static __device__ __forceinline__ T make(u32 payload, u32 fine, int) { return make_uint2(payload, fine); }
  static __device__ __forceinline__ u32 fine(const T& r, u32, int) { return r.y; }
  static __device__ __forceinline__ u32 payload(const T& r, int) { return r.x; }
};
struct Rec4 {   // ref << (fb + 1) | fine << 1 | sign;  needs ref < 2^(31 - fb)
  typedef u32 T;
  static __device__ __forceinline__ T make(u32 payload, u32 fine, int fb) { return ((payload & 0x7fffffffu) << (fb + 1)) | (fine << 1) | (payload >> 31); }
  static __device__ __forceinline__ u32 fine(const T& r, u32 fmask, int) { return (r >> 1) & fmask; }
  static __device__ __forceinline__ u32 payload(const T& r, int fb) { return (r >> (fb + 1)) | ((r & 1u) << 31); }
};
template <class REC, int C, int CL = COARSE_LOG>
__global__ __launch_bounds__(SORT2_THREADS) void k_coarse_scatter(const u32* __restrict__ scalars, size_t n, DigitLayout L, int key_shift,
                                                                   u32 fine_mask, int fb, const u32* __restrict__ binbase, int nwg,
                                                                   typename REC::T* __restrict__ tmp, const u32* __restrict__ bin_tot,
                                                                   u32* __restrict__ bin_cur, u32* __restrict__ bin_start_out) {
  constexpr int COARSE_BINS = 1 << CL;
  static_assert(COARSE_BINS <= SORT2_THREADS, "one lane per bin");
  __shared__ u32 cursor[COARSE_BINS];
  __shared__ u32 bscan[COARSE_BINS + 1];
  if (bin_tot) {      // scan-free form: binbase holds this kernel's RAW counts (k_coarse_count), the run of (bin, workgroup) is claimed
    bin_starts<COARSE_BINS>(bin_tot, bscan, bin_start_out);
    if (threadIdx.x < COARSE_BINS) {
      const u32 mine = binbase[(size_t)threadIdx.x * nwg + blockIdx.x];
      cursor[threadIdx.x] = bscan[threadIdx.x] + (mine ? atomicAdd(&bin_cur[threadIdx.x], mine) : 0u);
    }
  } else if (threadIdx.x < COARSE_BINS) {
    cursor[threadIdx.x] = binbase[(size_t)threadIdx.x * nwg + blockIdx.x];
  }
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * COARSE_PER_WG;
  const size_t hi = (lo + COARSE_PER_WG < n) ? lo + COARSE_PER_WG : n;
  u32 w[COARSE_PER_WG / SORT2_THREADS][8];
#pragma unroll
  for (int k = 0; k < COARSE_PER_WG / SORT2_THREADS; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * SORT2_THREADS;
    if (i < hi) load_scalar_canonical(scalars, i, w[k]);
  }
#pragma unroll
  for (int k = 0; k < COARSE_PER_WG / SORT2_THREADS; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * SORT2_THREADS;
    if (i < hi) {
      auto emit = [&](int, u32 key, u32 payload) {
        const u32 pos = counter_inc_agg(cursor, key >> key_shift);
        tmp[pos] = REC::make(payload, key & fine_mask, fb);
      };
      if constexpr (C != 0) walk_digits_merged<C>(w[k], L.table_stride, i, emit);
      else walk_digits(w[k], L, i, emit);
    }
  }
}
// The same with the records of 1024 scalars (one per lane, <= 15360 records) sorted by bin INSIDE LDS first and copied out as
// contiguous runs (~60 records = 480 bytes per bin and chunk): a what-if build with perfectly coalesced stores ran in 32 us
// against the 72 us of the isolated 8-byte stores above (profiles/r04g_*) -- the scatter, not the digits, is what that kernel
// waits for.  Per chunk: histogram by LDS atomics, one-wave exclusive scan, placement through LDS cursors, copy-out.
// Merged layout with a compile-time window width only (the digits are walked twice).
// records staged per chunk of 1024 scalars: NWIN <= 16 per scalar; with 1024 bins (two-byte bin tags) the 13 windows of the 20-bit layout
constexpr int stage_records(int C, int CL) { return (CL > 8 ? (254 / C + 1) : 16) * SORT2_THREADS; }
constexpr size_t stage_lds_bytes(int C, int CL, size_t rec_bytes) {
  return (size_t)stage_records(C, CL) * (rec_bytes + (CL > 8 ? 2 : 1)) + (size_t)(4 * (1 << CL) + 1) * 4;
}
constexpr int STAGE_RECORDS = 16 * SORT2_THREADS;
template <class REC, int C, int CL = COARSE_LOG>
__global__ __launch_bounds__(SORT2_THREADS) void k_coarse_scatter_staged(const u32* __restrict__ scalars, size_t n, size_t table_stride, int key_shift,
                                                                          u32 fine_mask, int fb, const u32* __restrict__ binbase, int nwg,
                                                                          typename REC::T* __restrict__ tmp, const u32* __restrict__ bin_tot,
                                                                          u32* __restrict__ bin_cur, u32* __restrict__ bin_start_out) {
  typedef typename REC::T R;
  constexpr int COARSE_BINS = 1 << CL;           // (shadows the 256 of the other kernels)
  constexpr int STAGE_RECORDS = stage_records(C, CL);
  constexpr int PER_LANE = COARSE_BINS / 64;     // bins per lane in the one-wave scan
  typedef typename std::conditional<(CL > 8), unsigned short, unsigned char>::type BinTag;
  static_assert(stage_lds_bytes(C, CL, sizeof(R)) <= (size_t)160 * 1024, "staging must fit the LDS");
  extern __shared__ __attribute__((aligned(16))) u32 lds_stage[];
  R* stage = reinterpret_cast<R*>(lds_stage);                                   // [STAGE_RECORDS]
  BinTag* stage_bin = reinterpret_cast<BinTag*>(stage + STAGE_RECORDS);         // [STAGE_RECORDS]
  u32* cursor = reinterpret_cast<u32*>(stage_bin + STAGE_RECORDS);              // [COARSE_BINS] running position of (bin, this workgroup) in tmp
  u32* hist = cursor + COARSE_BINS;                                             // [COARSE_BINS] records of the chunk per bin, then placement cursors
  u32* loff = hist + COARSE_BINS;                                               // [COARSE_BINS + 1] chunk-local exclusive offsets
  u32* gpos = loff + COARSE_BINS + 1;                                           // [COARSE_BINS] destination of the chunk's run of each bin
  static_assert(COARSE_BINS % 64 == 0 && COARSE_BINS <= SORT2_THREADS, "one wave scans PER_LANE bins per lane; one lane per bin elsewhere");
  const int tid = threadIdx.x;
  // scan-free form (bin_tot != null): cursor[] holds the bin STARTS and every chunk claims its run of a bin from the bin's global cursor
  if (bin_tot) bin_starts<COARSE_BINS>(bin_tot, loff, bin_start_out);      // (loff: COARSE_BINS + 1 words, free until the first chunk's scan)
  if (tid < COARSE_BINS) cursor[tid] = bin_tot ? loff[tid] : binbase[(size_t)tid * nwg + blockIdx.x];
  const size_t lo = (size_t)blockIdx.x * COARSE_PER_WG;
  const size_t hi = (lo + COARSE_PER_WG < n) ? lo + COARSE_PER_WG : n;
  u32 w[COARSE_PER_WG / SORT2_THREADS][8];
#pragma unroll
  for (int k = 0; k < COARSE_PER_WG / SORT2_THREADS; k++) {
    const size_t i = lo + tid + (size_t)k * SORT2_THREADS;
    if (i < hi) load_scalar_canonical(scalars, i, w[k]);
  }
#pragma unroll 1
  for (int k = 0; k < COARSE_PER_WG / SORT2_THREADS; k++) {
    const size_t i = lo + tid + (size_t)k * SORT2_THREADS;
    if (tid < COARSE_BINS) hist[tid] = 0;
    __syncthreads();
    if (i < hi) walk_digits_merged<C>(w[k], table_stride, i, [&](int, u32 key, u32) { counter_inc_agg(hist, key >> key_shift); });
    __syncthreads();
    if (tid < 64) {                      // exclusive scan of the bin counts by one wave: PER_LANE consecutive bins per lane
      u32 c[PER_LANE], sum = 0;
#pragma unroll
      for (int q = 0; q < PER_LANE; q++) { c[q] = hist[PER_LANE * tid + q]; sum += c[q]; }
      u32 incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const u32 up = (u32)__shfl_up((int)incl, d, 64);
        if (tid >= d) incl += up;
      }
      u32 run = incl - sum;
#pragma unroll
      for (int q = 0; q < PER_LANE; q++) { loff[PER_LANE * tid + q] = run; run += c[q]; }
      if (tid == 63) loff[COARSE_BINS] = incl;
    }
    __syncthreads();
    if (tid < COARSE_BINS) {
      const u32 cnt = hist[tid];
      if (bin_tot) {
        gpos[tid] = cursor[tid] + (cnt ? atomicAdd(&bin_cur[tid], cnt) : 0u);
      } else {
        gpos[tid] = cursor[tid];
        cursor[tid] += cnt;
      }
      hist[tid] = loff[tid];             // placement cursor
    }
    __syncthreads();
    if (i < hi) walk_digits_merged<C>(w[k], table_stride, i, [&](int, u32 key, u32 payload) {
      const u32 bin = key >> key_shift;
      const u32 pos = counter_inc_agg(hist, bin);
      stage[pos] = REC::make(payload, key & fine_mask, fb);
      stage_bin[pos] = (BinTag)bin;
    });
    __syncthreads();
    const u32 total = loff[COARSE_BINS];
    for (u32 p = tid; p < total; p += SORT2_THREADS) {
      const u32 b = stage_bin[p];
      tmp[gpos[b] + (p - loff[b])] = stage[p];
    }
    __syncthreads();
  }
}
// Which slice of which bin a fine workgroup handles.  A bin is the run [binbase[b nwg], binbase[(b + 1) nwg]) of the coarse-sorted
// records.  On uniform scalars every bin holds E / bins records and S workgroups each take an S-th of it; SHORT scalars do not fill
// the bins evenly -- bytes put every entry of a 17-bit window into bin 0, 248-bit coefficients put a fifteenth of all entries into the
// four bins of the 10-bit top window -- and with S slices for every bin two workgroups then counted and scattered a whole window alone
// (k_fine_scatter 50 -> 547 us at 2^20 bytes, 186 us at 248 bits).  So a bin gets max(S, ceil(len / cap)) slices, cap = 1.5 nominal
// slices: the uniform case keeps its S, an over-full bin is cut into as many workgroups as it needs.  Every workgroup derives the same
// plan from the bin boundaries (one load per lane and a block scan of two barriers): workgroup w -> (bin, slice); the histogram of bin b sits at
// finehist[F P_b + f S_b + s] (P_b = slices before bin b), one flat exclusive scan as before; workgroups past the last slice zero
// their block of the histogram, so the scan runs over the host's bound F * gridDim.
struct FineSlice { int bin, s, Sb; u32 lo, hi, start; bool active; };
// (sc, res: LDS scratch of the caller, SORT2_THREADS and 6 words -- no static LDS here: k_fine_scatter asks for all 160 KiB as dynamic)
__device__ __forceinline__ FineSlice fine_plan(const u32* __restrict__ binbase, int nwg, int bins, int S, u32 cap, u32* sc, u32* res) {
  const int tid = threadIdx.x;
  u32 start = 0, len = 0, Sb = 0;
  if (tid < bins) {
    start = binbase[(size_t)tid * nwg];
    len = binbase[(size_t)(tid + 1) * nwg] - start;
    Sb = (len + cap - 1) / cap;
    if (Sb < (u32)S) Sb = (u32)S;
  }
  if (tid == 0) res[0] = 0xffffffffu;
  // inclusive scan of Sb over the workgroup: inside each wave by shuffles, the 16 wave totals by the first wave -- two barriers
  const int lane = tid & 63, wave = tid >> 6;
  u32 v = Sb;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const u32 t = __shfl_up(v, off, 64); if (lane >= off) v += t; }
  if (lane == 63) sc[wave] = v;
  __syncthreads();
  if (wave == 0) {
    u32 t = (lane < SORT2_THREADS / 64) ? sc[lane] : 0u;
#pragma unroll
    for (int off = 1; off < SORT2_THREADS / 64; off <<= 1) { const u32 u = __shfl_up(t, off, 64); if (lane >= off) t += u; }
    if (lane < SORT2_THREADS / 64) sc[lane] = t;
  }
  __syncthreads();
  const u32 w = blockIdx.x, incl = v + (wave ? sc[wave - 1] : 0u), excl = incl - Sb;
  if (tid < bins && excl <= w && w < incl) { res[0] = (u32)tid; res[1] = w - excl; res[2] = Sb; res[3] = start; res[4] = len; res[5] = excl; }
  __syncthreads();
  FineSlice r;
  r.active = res[0] != 0xffffffffu;
  r.bin = (int)res[0]; r.s = (int)res[1]; r.Sb = (int)res[2];
  const u64 L = res[4];
  r.start = res[3];
  r.lo = res[3] + (u32)(L * (u64)r.s / (u64)r.Sb);
  r.hi = res[3] + (u32)(L * (u64)(r.s + 1) / (u64)r.Sb);
  return r;
}
constexpr int FINE_UNROLL = 8;
constexpr int FINE_MAX = 8192;     // buckets per bin: NB / 256 (128 merged c = 16, 2048 generic c = 16, 8192 merged c = 22)
template <class REC>
__global__ __launch_bounds__(SORT2_THREADS) void k_fine_count(const typename REC::T* __restrict__ tmp, const u32* __restrict__ binbase, int nwg, int F,
                                                               int S, int fb, u32* __restrict__ finehist, int bins, u32 cap, u32* __restrict__ bucket_tot) {
  __shared__ u32 hist[FINE_MAX];
  __shared__ u32 sc[SORT2_THREADS];
  const FineSlice p = fine_plan(binbase, nwg, bins, S, cap, sc, hist);
  __syncthreads();                      // (hist is the plan's scratch until here)
  if (!p.active) {                      // past the last slice: this block of the histogram scans as zeros
    for (int f = threadIdx.x; f < F; f += SORT2_THREADS) finehist[(size_t)blockIdx.x * F + f] = 0;
    return;
  }
  for (int f = threadIdx.x; f < F; f += SORT2_THREADS) hist[f] = 0;
  __syncthreads();
  const u32 lo = p.lo, hi = p.hi;
  // FINE_UNROLL independent loads in flight per lane: the loop is otherwise one global-load latency per entry
  for (u32 base = lo + threadIdx.x; base < hi; base += FINE_UNROLL * SORT2_THREADS) {
    u32 f[FINE_UNROLL];
#pragma unroll
    for (int k = 0; k < FINE_UNROLL; k++) {
      const u32 e = base + k * SORT2_THREADS;
      f[k] = (e < hi) ? REC::fine(tmp[e], (u32)F - 1u, fb) : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < FINE_UNROLL; k++) if (f[k] != 0xffffffffu) counter_inc_agg(hist, f[k]);
  }
  __syncthreads();
  const size_t hbase = (size_t)F * (blockIdx.x - (u32)p.s);        // F P_b: the slices of a bin are consecutive workgroups
  for (int f = threadIdx.x; f < F; f += SORT2_THREADS) {
    const u32 v = hist[f];
    finehist[hbase + (size_t)f * p.Sb + p.s] = v;
    if (bucket_tot && v) atomicAdd(&bucket_tot[(size_t)p.bin * F + f], v);      // the bucket's size over all slices (k_fine_scatter: no global scan)
  }
}
// Direct form: every record is stored straight to its final position (isolated 4-byte stores).  Only used when a
// bin has more buckets than the staged kernel below has LDS for.
template <class REC>
__global__ __launch_bounds__(SORT2_THREADS) void k_fine_scatter_direct(const typename REC::T* __restrict__ tmp, const u32* __restrict__ binbase, int nwg, int F,
                                                                 int S, int fb, const u32* __restrict__ finebase, u32* __restrict__ offsets,
                                                                 u32* __restrict__ entries, size_t nbuckets, int bins, u32 cap) {
  __shared__ u32 cursor[FINE_MAX];
  __shared__ u32 sc[SORT2_THREADS];
  const FineSlice p = fine_plan(binbase, nwg, bins, S, cap, sc, cursor);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[nbuckets] = finebase[(size_t)F * gridDim.x];   // total entries
  if (!p.active) return;
  const size_t hbase = (size_t)F * (blockIdx.x - (u32)p.s);
  for (int f = threadIdx.x; f < F; f += SORT2_THREADS) {
    const u32 base = finebase[hbase + (size_t)f * p.Sb + p.s];
    cursor[f] = base;
    if (p.s == 0) offsets[(size_t)p.bin * F + f] = base;      // start of bucket (bin, f)
  }
  __syncthreads();
  const u32 lo = p.lo, hi = p.hi;
  for (u32 base = lo + threadIdx.x; base < hi; base += FINE_UNROLL * SORT2_THREADS) {
    typename REC::T r[FINE_UNROLL];
#pragma unroll
    for (int k = 0; k < FINE_UNROLL; k++) {
      const u32 e = base + k * SORT2_THREADS;
      if (e < hi) r[k] = tmp[e];
    }
#pragma unroll
    for (int k = 0; k < FINE_UNROLL; k++)
      if (base + k * SORT2_THREADS < hi) entries[counter_inc_agg(cursor, REC::fine(r[k], (u32)F - 1u, fb))] = REC::payload(r[k], fb);
  }
}

// Staged form: the workgroup sorts up to STAGE_CAP records of its slice by bucket INSIDE LDS (count, prefix, place),
// then copies the staged run out with consecutive lanes writing consecutive entries -- bucket runs leave as
// contiguous stores instead of one dirty sector per record (HBM write traffic of this kernel 3.5x -> ~1x payload).
#ifndef MZK_STAGE_CAP
#define MZK_STAGE_CAP 8192
#endif
// 8192 records per round = 56 KB of LDS and 8 records in registers per lane: two workgroups per CU, whose phases overlap (16384 =
// one workgroup per CU: sort 0.170 -> 0.163 ms at 2^20, 0.726 -> 0.668 at 2^22; 4096 loses on the generic layout: profiles/r04l_*)
constexpr int STAGE_CAP = MZK_STAGE_CAP;
constexpr int STAGE_PER_LANE = STAGE_CAP / SORT2_THREADS;     // records per lane per round
constexpr int STAGE_F_MAX = 2048;
template <class REC>
__global__ __launch_bounds__(SORT2_THREADS) void k_fine_scatter(const typename REC::T* __restrict__ tmp, const u32* __restrict__ binbase, int nwg, int F,
                                                                 int S, int fb, const u32* __restrict__ finebase, u32* __restrict__ offsets,
                                                                 u32* __restrict__ entries, size_t nbuckets, int bins, u32 cap, const u32* __restrict__ bucket_tot,
                                                                 u32* __restrict__ bucket_cur) {
  extern __shared__ u32 lds_fs[];
  u32* gbase = lds_fs;                 // [F]   global position of the next entry of bucket f written by this workgroup
  u32* cnt = gbase + F;                // [F]   records of bucket f in the current round
  u32* lpre = cnt + F;                 // [F]   exclusive prefix of cnt
  u32* scan = lpre + F;                // [SORT2_THREADS] block-scan scratch
  u32* spay = scan + SORT2_THREADS;    // [STAGE_CAP] staged payloads, bucket-sorted
  unsigned short* skey = reinterpret_cast<unsigned short*>(spay + STAGE_CAP);   // [STAGE_CAP] their buckets
  const int tid = threadIdx.x;
  const FineSlice p = fine_plan(binbase, nwg, bins, S, cap, scan, gbase);
  __syncthreads();                     // (gbase is the plan's scratch until here)
  if (blockIdx.x == 0 && tid == 0) offsets[nbuckets] = bucket_tot ? binbase[(size_t)bins * nwg] : finebase[(size_t)F * gridDim.x];   // total entries
  if (!p.active) return;
  const size_t hbase = (size_t)F * (blockIdx.x - (u32)p.s);
  if (bucket_tot) {
    // No global scan ran.  The bin's records start at a known place (the coarse sort's boundary) and hold its F buckets one after the
    // other: bucket offsets = bin start + exclusive prefix of the bin's F bucket totals (k_fine_count summed them with atomics), and this
    // slice's run inside a bucket is claimed from the bucket's cursor (the order of the slices inside a bucket is arbitrary, as the order
    // of entries inside a bucket always was).  Whatever the number of slices a skewed bin was cut into, a workgroup reads F totals.
    const int per2 = (F + SORT2_THREADS - 1) / SORT2_THREADS;
    u32 local = 0;
    for (int q = 0; q < per2; q++) { const int f = tid * per2 + q; if (f < F) { const u32 v = bucket_tot[(size_t)p.bin * F + f]; cnt[f] = v; local += v; } }
    u32 run = p.start + block_excl_scan<SORT2_THREADS>(local, scan, nullptr);
    for (int q = 0; q < per2; q++) {
      const int f = tid * per2 + q;
      if (f < F) {
        const u32 v = cnt[f];
        const u32 mine = finebase[hbase + (size_t)f * p.Sb + p.s];
        gbase[f] = run + (mine ? atomicAdd(&bucket_cur[(size_t)p.bin * F + f], mine) : 0u);
        if (p.s == 0) offsets[(size_t)p.bin * F + f] = run;
        run += v;
      }
    }
    __syncthreads();                                            // (cnt is the rounds' counter array from here on)
  } else {
    for (int f = tid; f < F; f += SORT2_THREADS) {
      const u32 base = finebase[hbase + (size_t)f * p.Sb + p.s];
      gbase[f] = base;
      if (p.s == 0) offsets[(size_t)p.bin * F + f] = base;      // start of bucket (bin, f)
    }
  }
  const u32 lo = p.lo, hi = p.hi;
  const int per = (F + SORT2_THREADS - 1) / SORT2_THREADS;     // counters per lane in the prefix (<= 2)
  for (u32 chunk = lo; chunk < hi; chunk += STAGE_CAP) {
    const u32 cend = (hi - chunk > (u32)STAGE_CAP) ? chunk + STAGE_CAP : hi;
    for (int f = tid; f < F; f += SORT2_THREADS) cnt[f] = 0;
    __syncthreads();
    typename REC::T r[STAGE_PER_LANE];
    u32 rk[STAGE_PER_LANE];
    const u32 fmask = (u32)F - 1u;
#pragma unroll
    for (int k = 0; k < STAGE_PER_LANE; k++) {
      const u32 e = chunk + tid + k * SORT2_THREADS;
      if (e < cend) r[k] = tmp[e];
    }
#pragma unroll
    for (int k = 0; k < STAGE_PER_LANE; k++) rk[k] = (chunk + tid + k * SORT2_THREADS < cend) ? counter_inc_agg(cnt, REC::fine(r[k], fmask, fb)) : 0u;
    __syncthreads();
    // exclusive prefix of cnt[0..F): lane-local run of `per` counters, block scan over the lane sums
    u32 local = 0;
    for (int q = 0; q < per; q++) { const int f = tid * per + q; if (f < F) local += cnt[f]; }
    u32 run = block_excl_scan<SORT2_THREADS>(local, scan, nullptr);
    for (int q = 0; q < per; q++) { const int f = tid * per + q; if (f < F) { lpre[f] = run; run += cnt[f]; } }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < STAGE_PER_LANE; k++) {
      if (chunk + tid + k * SORT2_THREADS < cend) {
        const u32 fk = REC::fine(r[k], fmask, fb);
        const u32 pos = lpre[fk] + rk[k];
        spay[pos] = REC::payload(r[k], fb);
        skey[pos] = (unsigned short)fk;
      }
    }
    __syncthreads();
    const u32 m = cend - chunk;
    for (u32 j = tid; j < m; j += SORT2_THREADS) {
      const u32 f = skey[j];
      entries[gbase[f] + (j - lpre[f])] = spay[j];
    }
    __syncthreads();
    for (int f = tid; f < F; f += SORT2_THREADS) gbase[f] += cnt[f];
    __syncthreads();
  }
}

// ---- 2. exclusive scan (three small kernels) -----------------------------------------------------------
constexpr int SCAN_ITEMS = 8;                      // per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;       // 2048 per block
__global__ __launch_bounds__(256) void k_scan_local(const u32* __restrict__ in, u32* __restrict__ out, u32* __restrict__ block_sums, size_t n) {
  __shared__ u32 sh[256];
  const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ITEMS;
  u32 v[SCAN_ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    v[k] = (base + k < n) ? in[base + k] : 0u;
    sum += v[k];
  }
  u32 block_total;
  u32 excl = block_excl_scan<256>(sum, sh, &block_total);
  if (threadIdx.x == 255) block_sums[blockIdx.x] = block_total;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < n) out[base + k] = excl;
    excl += v[k];
  }
}
// offsets[i] += sum of the block totals in front of block(i); offsets[n] = total.  Every workgroup (256 consecutive i: ONE block)
// sums the totals in front of its block itself -- a few dozen values -- instead of a launch of its own for their scan.
__global__ __launch_bounds__(256) void k_scan_finish(u32* __restrict__ offsets, const u32* __restrict__ block_sums, size_t nblocks, size_t n) {
  __shared__ u32 part[4];
  static_assert(SCAN_BLOCK % 256 == 0, "one block index per workgroup");
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t first = (size_t)blockIdx.x * blockDim.x;
  const bool total_wg = first >= n;                                   // the one workgroup past the data writes offsets[n]
  const size_t blk = total_wg ? nblocks : first / SCAN_BLOCK;
  u32 acc = 0;
  for (size_t j = threadIdx.x; j < blk; j += 256) acc += block_sums[j];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += (u32)__shfl_xor((int)acc, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  const u32 pre = part[0] + part[1] + part[2] + part[3];
  if (total_wg) { if (threadIdx.x == 0) offsets[n] = pre; }
  else if (i < n) offsets[i] = offsets[i] + pre;
}

// offsets[i] += prefix[block(i)] for a prefix array that is already scanned (prefix[nblocks] = total -> offsets[n])
__global__ __launch_bounds__(256) void k_scan_add(u32* __restrict__ offsets, const u32* __restrict__ prefix, size_t nblocks, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) offsets[i] += prefix[i / SCAN_BLOCK];
  else if (i == n) offsets[n] = prefix[nblocks];
}
// out[i] = sum_{j<i} in[j] for i <= n (out[n] = total); in == out allowed; scratch: scan_scratch_words(n) words.
// Two launches: block-local scans, then every workgroup adds the totals in front of its block (k_scan_finish re-sums them
// itself: a few dozen to a few hundred values).  (A single-workgroup scan of the 65536 counters was measured: 40 us SLOWER per
// scan than the launches -- one CU cannot stream and shuffle-scan 256 KiB as fast as 32 workgroups do, launch latency included.)
// Above SCAN_DIRECT_BLOCKS block totals that re-summing would be quadratic (the generic layout at 2^24 has 16384 of them, 22-bit
// windows 65536: ~10^9 .. 10^10 redundant loads), so the totals are scanned by a recursive call and added (four or five launches).
constexpr size_t SCAN_DIRECT_BLOCKS = 2048;
static size_t scan_scratch_words(size_t n) {
  const size_t sb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  return sb + 2 + (sb > SCAN_DIRECT_BLOCKS ? scan_scratch_words(sb) : 0);
}
static int launch_exclusive_scan(const u32* in, u32* out, size_t n, u32* scratch, hipStream_t s) {
  const size_t sb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  hipLaunchKernelGGL(k_scan_local, dim3((unsigned)sb), dim3(256), 0, s, in, out, scratch, n);
  if (sb <= SCAN_DIRECT_BLOCKS) {
    hipLaunchKernelGGL(k_scan_finish, dim3((unsigned)((n + 255) / 256 + 1)), dim3(256), 0, s, out, (const u32*)scratch, sb, n);
  } else {
    MZK_TRY(launch_exclusive_scan((const u32*)scratch, scratch, sb, scratch + sb + 2, s));      // scratch[b] = totals in front of block b, scratch[sb] = total
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)(n / 256 + 1)), dim3(256), 0, s, out, (const u32*)scratch, sb, n);
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// ---- 4. bucket accumulation, segmented ------------------------------------------------------------------
// The sorted entry array is cut into fixed segments of `seg` entries, one lane per segment, so the work
// per lane is equal whatever the scalar distribution (no Poisson tail, no skew cliff).  A lane walks
// its segment, keeps one XYZZ accumulator, and flushes it whenever the bucket changes.  The partial
// of (segment t, bucket b) goes to slot t + b: (t, b) pairs met in order strictly increase t + b, so
// slots are unique, and all partials of bucket b sit in the contiguous slot range
// [offsets[b] / seg + b, (offsets[b+1] - 1) / seg + b] -- pass 2 sums that range (every slot of that range is
// written: segment t overlaps bucket b exactly when slot t + b lies in it).
// `seg` is chosen by the host so that the grid is ONE full round of resident waves (see accumulate_segment): the
// kernel is a long dependent loop per lane, so a partial second round would run at a fraction of the occupancy.
//
// PREFETCH = true keeps the next entry's point in 16 extra registers (139 VGPRs -> 3 waves per SIMD); false loads the
// point where it is used (123 VGPRs -> 4 waves per SIMD) and leaves the latency to the other three waves.
// SENT (the one-kernel sort of the grid-batched commitments): a polynomial's entry region has a fixed capacity and its unused tail
// is filled with MANY_SENTINEL entries, which are skipped without touching the table.
constexpr u32 MANY_SENTINEL = 0xffffffffu;
// Segment length as the kernels see it.  The host sizes `seg` for the MOST entries the scalars can have (every digit non-zero); short
// or sparse scalars -- bits, bytes, 64-bit values, half of them zero -- emit a fraction of that, and with the host's length the
// entries would fill the first few workgroups' lanes with full-length chains while the other CUs idle (2^20 16-bit scalars: 1/15 of
// the entries, accumulate 0.32 ms against 1.1 on full-width ones).  With t_max != 0 every kernel of the segment stage derives the
// length from the entry count the sort left at offsets[nbuckets]: the entries spread over all t_max lanes again (never more: slot
// t + b stays inside the t_max + nbuckets slots), down to SEG_MIN entries per lane.  Full-width uniform scalars get the host's value.
constexpr u32 SEG_MIN = 8;
__device__ __forceinline__ u32 segment_length(u32 seg_host, u32 t_max, u32 total_entries) {
  if (t_max == 0) return seg_host;
  const u32 v = (total_entries + t_max - 1) / t_max;
  return v < SEG_MIN ? SEG_MIN : v;
}
template <bool PREFETCH, bool SENT = false>
__global__ __launch_bounds__(256) void k_seg_accumulate(const u32* __restrict__ points_mont, const u32* __restrict__ offsets,
                                                         const u32* __restrict__ entries, u32* __restrict__ slots, size_t nbuckets,
                                                         u32 seg_host, u32 t_max) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 total_entries = offsets[nbuckets];
  const u32 seg = segment_length(seg_host, t_max, total_entries);
  const u64 e0w = (u64)t * seg;
  if (e0w >= total_entries) return;
  const u32 e0 = (u32)e0w;
  const u32 e1 = (e0w + seg < total_entries) ? e0 + seg : total_entries;
  // bucket of entry e0: last b with offsets[b] <= e0
  size_t lo = 0, hi = nbuckets;  // invariant: offsets[lo] <= e0 < offsets[hi] (offsets[nbuckets] = total)
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= e0) lo = mid; else hi = mid;
  }
  size_t b = lo;
  u32 bend = offsets[b + 1];
  Xyzz acc = xyzz_inf();
  u32 ent_next = entries[e0];
  u32 wn[16];
  if (PREFETCH) {
    // software pipeline: the point of entry e+1 is requested before the ~3000-instruction madd of entry e,
    // so the dependent entries[] -> points[] gather is in flight behind arithmetic instead of in front of it
    const size_t idx = ent_next & 0x7fffffffu;
    load_words8(points_mont + idx * 16, wn);
    load_words8(points_mont + idx * 16 + 8, wn + 8);
  }
  for (u32 e = e0; e < e1; e++) {
    const u32 ent = ent_next;
    u32 w[16];
    if (PREFETCH) {
#pragma unroll
      for (int k = 0; k < 16; k++) w[k] = wn[k];
    } else {
      const size_t idx = (SENT && ent == MANY_SENTINEL) ? 0 : (ent & 0x7fffffffu);
      load_words8(points_mont + idx * 16, w);
      load_words8(points_mont + idx * 16 + 8, w + 8);
    }
    if (e + 1 < e1) {
      ent_next = entries[e + 1];
      if (PREFETCH) {
        const size_t idx = ent_next & 0x7fffffffu;
        load_words8(points_mont + idx * 16, wn);
        load_words8(points_mont + idx * 16 + 8, wn + 8);
      }
    }
    if (e >= bend) {
      xyzz_gstore_raw(slots, t + b, acc);
      acc = xyzz_inf();
      do { b++; bend = offsets[b + 1]; } while (e >= bend);
    }
    if (affine_words_is_inf(w)) continue;  // infinity contributes nothing (curve.rs:107-109)
    if (SENT && ent == MANY_SENTINEL) continue;
    acc = xyzz_madd_signed_with<FeAsm>(acc, affine_load_mont(w), (ent >> 31) != 0);
  }
  xyzz_gstore_raw(slots, t + b, acc);
}
// pass 2: buckets[b] = sum of slots [offsets[b] / seg + b, (offsets[b+1] - 1) / seg + b]   (inclusive end:
// the last entry of bucket b is offsets[b+1]-1)
// Buckets with more than HEAVY_SLOTS partials (skewed scalars: bit vectors, repeated values) would be one long
// serial chain; they are queued in `heavy` (count at heavy[0], then (bucket id, end of entries) pairs) and summed by a whole
// workgroup each (k_seg_combine_heavy).
// The threshold is 32 partials; 16 for the one-kernel sort of the grid-batched pass (HEAVY_SLOTS_SORT1): there a bucket has 8 - 9
// partials and the signed digits of SHORT coefficients (31-byte chunks) put the carry out of their last non-zero window into bucket 0
// of the window above -- ONE bucket per polynomial with three times the entries of the others, a chain of ~30 in a kernel that is as
// long as its longest chain (256 x 2^10: segment combine 0.149 -> 0.105 ms).  Elsewhere 16 loses: 64 x 2^12 full-width coefficients have
// ~12 such buckets per polynomial (the 4-bit top window), 768 deferred buckets are two rounds of the workgroup kernel: 0.114 -> 0.168.
constexpr u32 HEAVY_SLOTS = 32;
constexpr u32 HEAVY_SLOTS_SORT1 = 16;
// end of bucket b's entries.  The one-kernel sort of the grid-batched commitments (k_many_sort1) leaves a tail of sentinels behind the
// LAST bucket of every polynomial's fixed-capacity region; `tails` (one word per polynomial, or null) is where the real entries end,
// so that the sentinel-only segments are not summed as a chain of identity partials (short coefficients -- the 31-byte chunks of the
// reference's DAS callers -- leave ~1100 sentinels per 1024-coefficient polynomial: nine extra dependent additions in a kernel
// that is as long as its longest chain, 73 -> 148 us at 256 x 2^10)
__device__ __forceinline__ u32 bucket_end(const u32* __restrict__ offsets, const u32* __restrict__ tails, int lg_nb, size_t b) {
  if (tails && (b & (((size_t)1 << lg_nb) - 1)) == (((size_t)1 << lg_nb) - 1)) return tails[b >> lg_nb];
  return offsets[b + 1];
}
// Layout of `heavy`: [0] count, [1] the segment length in force (written by the combine kernel), [2, 2 + HEAVY_GRID) arrival counters of k_seg_combine_heavy's shared buckets (zeroed with
// the count, one memset), then (bucket id, end of its entries as the deferring kernel saw it -- bucket_end) pairs, then HEAVY_GRID
// XYZZ records of scratch for the workgroups that share a bucket.
// Behind the arrival counters, inside the same cleared header: the scan-free sort's per-bin totals and cursors (k_coarse_count /
// k_coarse_scatter*: SORT_CTR_BINS words each) -- one memset per call clears everything that has to start at zero.
constexpr int HEAVY_GRID = 512;
constexpr int SORT_CTR_BINS = 1024;               // the most coarse bins any layout uses (20-bit merged windows)
constexpr int SORT_CTR_AT = 2 + HEAVY_GRID;       // bin totals at heavy[SORT_CTR_AT ..), bin cursors SORT_CTR_BINS words further
constexpr int HEAVY_HDR = SORT_CTR_AT + 2 * SORT_CTR_BINS;
constexpr size_t HEAVY_CLEAR_BYTES = (size_t)HEAVY_HDR * 4;
// (a multiple of four words: the scratch records behind the list are read and written as uint4)
__host__ __device__ constexpr size_t heavy_list_words(size_t max_heavy) { return (HEAVY_HDR + 2 * max_heavy + 2 + 3) & ~(size_t)3; }
__host__ __device__ constexpr size_t heavy_total_words(size_t max_heavy) { return heavy_list_words(max_heavy) + (size_t)HEAVY_GRID * 32 + 8; }
__device__ __forceinline__ bool defer_heavy(size_t b, size_t s0, size_t s1, u32 o1, u32* __restrict__ heavy, bool leader, u32 threshold = HEAVY_SLOTS) {
  if (s1 - s0 + 1 <= threshold) return false;
  if (leader) { const u32 i = atomicAdd(&heavy[0], 1u); heavy[HEAVY_HDR + 2 * i] = (u32)b; heavy[HEAVY_HDR + 2 * i + 1] = o1; }
  return true;
}
// one lane per bucket: with 2^19 buckets of one or two partials each (generic layout) the kernel is bound by
// the record traffic, not by a dependent chain
__global__ __launch_bounds__(128) void k_seg_combine_wide(const u32* __restrict__ slots, const u32* __restrict__ offsets,
                                                           u32* __restrict__ buckets, size_t nbuckets, u32 seg_host, u32* __restrict__ heavy, u32 t_max) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  const u32 seg = segment_length(seg_host, t_max, offsets[nbuckets]);
  if (b == 0) heavy[1] = seg;                        // k_seg_combine_heavy reads it there
  const u32 o0 = offsets[b], o1 = offsets[b + 1];
  Xyzz acc = xyzz_inf();
  if (o1 > o0) {
    const size_t s0 = (size_t)(o0 / seg) + b, s1 = (size_t)((o1 - 1) / seg) + b;
    if (defer_heavy(b, s0, s1, o1, heavy, true)) return;
    for (size_t sl = s0; sl <= s1; sl++) acc = xyzz_add_with<FeAsm>(acc, xyzz_gload_raw(slots, sl));
  }
  xyzz_gstore(buckets, b, acc);
}
// the same point held by the lane `dist` lanes away (every lane of a quad holds its quad's whole point)
__device__ __forceinline__ Xyzz xyzz_from_lane_xor(const Xyzz& p, int dist) {
  Xyzz r;
#pragma unroll
  for (int i = 0; i < FqParams::L; i++) {
    r.X.l[i] = (u32)__shfl_xor((int)p.X.l[i], dist, 64); r.Y.l[i] = (u32)__shfl_xor((int)p.Y.l[i], dist, 64);
    r.ZZ.l[i] = (u32)__shfl_xor((int)p.ZZ.l[i], dist, 64); r.ZZZ.l[i] = (u32)__shfl_xor((int)p.ZZZ.l[i], dist, 64);
  }
  return r;
}
// One DPP quad per bucket (QPB = 1: the shipped form).  A bucket's partials form a serial chain of additions, so the quad-cooperative
// addition cuts the kernel's latency (the quad also splits the 128-byte records).  QPB = 2 / 4 -- adjacent quads take every QPB-th
// partial each and fold their sums through shuffles, chain of 9 -> 5 + 1 / 3 + 2 in a grid-batched pass of 64 x 2^12 coefficients --
// exist in the tuning build only: measured slower everywhere (see msm_many_dev_impl), the kernel is bound by its instruction count.
template <int QPB>
__global__ __launch_bounds__(128) void k_seg_combine(const u32* __restrict__ slots, const u32* __restrict__ offsets, u32* __restrict__ buckets,
                                                      size_t nbuckets, u32 seg_host, u32* __restrict__ heavy, const u32* __restrict__ tails, int lg_nb, u32 t_max) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t b = tid / (4 * QPB);
  const int lane = (int)(tid & 3), q = (int)((tid >> 2) % QPB);
  if (b >= nbuckets) return;                         // (a bucket's QPB quads are adjacent lanes of one wave: uniform for the shuffles below)
  const u32 seg = segment_length(seg_host, t_max, offsets[nbuckets]);
  if (tid == 0) heavy[1] = seg;                      // k_seg_combine_heavy reads it there
  const u32 o0 = offsets[b], o1 = bucket_end(offsets, tails, lg_nb, b);
  Xyzz acc = xyzz_inf();
  if (o1 > o0) {
    const size_t s0 = (size_t)(o0 / seg) + b, s1 = (size_t)((o1 - 1) / seg) + b;
    if (defer_heavy(b, s0, s1, o1, heavy, lane == 0 && q == 0, tails ? HEAVY_SLOTS_SORT1 : HEAVY_SLOTS)) return;       // uniform over the bucket's quads
    if (QPB == 1) {
      acc = xyzz_gload_raw_quad(slots, s0, lane);
      for (size_t sl = s0 + 1; sl <= s1; sl++) acc = xyzz_add_quad(acc, xyzz_gload_raw_quad(slots, sl, lane), lane);
    } else {
      for (size_t sl = s0 + q; sl <= s1; sl += QPB) acc = xyzz_add_quad(acc, xyzz_gload_raw_quad(slots, sl, lane), lane);
    }
  }
  if (QPB > 1) {
    // (all lanes of the bucket's quads are here, empty buckets included: the shuffles see active partners)
    acc = xyzz_add_quad(acc, xyzz_from_lane_xor(acc, 4), lane);
    if (QPB > 2) acc = xyzz_add_quad(acc, xyzz_from_lane_xor(acc, 8), lane);
    if (q != 0) return;
  }
  xyzz_gstore_quad(buckets, b, acc, lane);
}
constexpr int HEAVY_THREADS = 256;
constexpr int HEAVY_QUADS = HEAVY_THREADS / 4;
// the workgroup's HEAVY_QUADS partial sums (one per quad, in sh) -> sh[0]
__device__ __forceinline__ void heavy_tree(u32* sh, const Xyzz& acc, int quad, int lane) {
  xyzz_gstore_quad(sh, quad, acc, lane);
  __syncthreads();
  for (int off = HEAVY_QUADS / 2; off >= 1; off >>= 1) {
    if (quad < off) {
      const Xyzz x = xyzz_gload_quad(sh, quad, lane), y = xyzz_gload_quad(sh, quad + off, lane);
      xyzz_gstore_quad(sh, quad, xyzz_add_quad(x, y, lane), lane);
    }
    __syncthreads();
  }
}
// Many heavy buckets: one workgroup each, in turn.  FEW of them (at most half the grid: bit vectors, constant polynomials, all-equal
// scalars -- one to fifteen buckets holding up to 2^20 entries, i.e. ~17000 partials each): G = grid / count workgroups share a bucket,
// each sums a contiguous G-th of its partials into a scratch record, and the last one to arrive (a counter per bucket) sums the G
// records.  One workgroup per bucket made a commit to 2^20 ones 2.0 ms and to 2^20 equal scalars 5.1 ms against 1.5 ms on uniform ones.
__global__ __launch_bounds__(HEAVY_THREADS) void k_seg_combine_heavy(const u32* __restrict__ slots, const u32* __restrict__ offsets,
                                                                      u32* __restrict__ buckets, u32* __restrict__ heavy, size_t max_heavy) {
  __shared__ __attribute__((aligned(16))) u32 sh[HEAVY_QUADS * 32];
  __shared__ u32 last_arrival;
  const u32 count = heavy[0];
  if (count == 0) return;
  const u32 seg = heavy[1];                          // the segment length the deferring kernel used (segment_length)
  const int lane = threadIdx.x & 3, quad = threadIdx.x >> 2;
  const u32* list = heavy + HEAVY_HDR;
  if (2 * count > gridDim.x) {
    for (u32 h = blockIdx.x; h < count; h += gridDim.x) {
      const size_t b = list[2 * h];
      const u32 o0 = offsets[b], o1 = list[2 * h + 1];
      const size_t s0 = (size_t)(o0 / seg) + b, s1 = (size_t)((o1 - 1) / seg) + b;
      Xyzz acc = xyzz_inf();
      for (size_t sl = s0 + quad; sl <= s1; sl += HEAVY_QUADS) acc = xyzz_add_quad(acc, xyzz_gload_raw_quad(slots, sl, lane), lane);
      heavy_tree(sh, acc, quad, lane);
      if (quad == 0) xyzz_gstore_quad(buckets, b, xyzz_gload_quad(sh, 0, lane), lane);
      __syncthreads();
    }
    return;
  }
  const u32 G = gridDim.x / count;
  const u32 h = blockIdx.x / G, part = blockIdx.x % G;
  if (h >= count) return;
  u32* scratch = heavy + heavy_list_words(max_heavy);          // record h G + part
  const size_t b = list[2 * h];
  const u32 o0 = offsets[b], o1 = list[2 * h + 1];
  const size_t s0 = (size_t)(o0 / seg) + b, s1 = (size_t)((o1 - 1) / seg) + b;
  if (s1 - s0 + 1 <= 16 * HEAVY_QUADS) {                        // up to 16 partials per quad: sharing costs a second tree (six dependent
    if (part != 0) return;                                      // additions) and two fences -- 256 buckets of 24 partials 40 -> 90 us, 48
                                                                // buckets of 260 partials 40 -> 75 us -- so part 0 takes these alone
    Xyzz acc = xyzz_inf();
    for (size_t sl = s0 + quad; sl <= s1; sl += HEAVY_QUADS) acc = xyzz_add_quad(acc, xyzz_gload_raw_quad(slots, sl, lane), lane);
    heavy_tree(sh, acc, quad, lane);
    if (quad == 0) xyzz_gstore_quad(buckets, b, xyzz_gload_quad(sh, 0, lane), lane);
    return;
  }
  const size_t per = (s1 - s0 + G) / G;                         // ceil(partials / G)
  const size_t lo = s0 + (size_t)part * per;
  size_t hi = lo + per - 1;
  if (hi > s1) hi = s1;
  Xyzz acc = xyzz_inf();
  for (size_t sl = lo + quad; sl <= hi; sl += HEAVY_QUADS) acc = xyzz_add_quad(acc, xyzz_gload_raw_quad(slots, sl, lane), lane);
  heavy_tree(sh, acc, quad, lane);
  if (quad == 0) xyzz_gstore_quad(scratch, (size_t)h * G + part, xyzz_gload_quad(sh, 0, lane), lane);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last_arrival = (atomicAdd(&heavy[2 + h], 1u) == G - 1) ? 1u : 0u;
  __syncthreads();
  if (!last_arrival) return;
  __threadfence();
  acc = xyzz_inf();
  for (u32 q = quad; q < G; q += HEAVY_QUADS) acc = xyzz_add_quad(acc, xyzz_gload_quad(scratch, (size_t)h * G + q, lane), lane);
  heavy_tree(sh, acc, quad, lane);
  if (quad == 0) xyzz_gstore_quad(buckets, b, xyzz_gload_quad(sh, 0, lane), lane);
}

// ---- 5. bucket reduction: sum_b (b+1) B_b per bucket set, by in-place halving ---------------------------
// Write b in binary.  Step t adds the upper half of every live block onto its lower half:
//   main block  M (size B / 2^t at offset 0):      M[j] += M[j + h]      -> still "all buckets, folded"
//   the upper half it just consumed stays in place and becomes a new live block U_t (the buckets whose
//   bit (lgB-1-t) is set), which later steps keep folding onto its own lower half.
// After lgB steps  buf[0] = sum_b B_b  and  buf[2^j] = sum_{b : bit j of b set} B_b, hence
//   sum_b (b + 1) B_b = buf[0] + sum_j 2^j buf[2^j].
// 2 B additions in total (the same as the serial running-sum trick) but only lgB dependent steps.
// Every addition is done by a DPP quad (mzk_coop.h): these steps are latency chains, not throughput work.
__device__ __forceinline__ void halve_op(u32* __restrict__ buf, int lgB, int t, size_t id, int lane) {
  const int lgh = lgB - t - 1;                    // log2(half)
  const size_t a = id >> lgh, j = id & (((size_t)1 << lgh) - 1);
  const size_t base = (a == 0) ? 0 : ((size_t)1 << (lgB - a));
  const size_t idx = base + j;
  const Xyzz x = xyzz_gload_quad(buf, idx, lane), y = xyzz_gload_quad(buf, idx + ((size_t)1 << lgh), lane);
  xyzz_gstore_quad(buf, idx, xyzz_add_quad(x, y, lane), lane);
}
// one lane per addition: for the early steps of MANY bucket sets (generic layout: 16 sets x 16 Ki additions) the
// work is throughput-bound and the plain addition issues fewer instructions than the quad form
__global__ __launch_bounds__(128) void k_halve_step_wide(u32* __restrict__ buckets, int lgB, int t) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)(t + 1) << (lgB - t - 1);
  if (id >= total) return;
  u32* buf = buckets + ((size_t)blockIdx.y << lgB) * 32;
  const int lgh = lgB - t - 1;
  const size_t a = id >> lgh, j = id & (((size_t)1 << lgh) - 1);
  const size_t idx = ((a == 0) ? 0 : ((size_t)1 << (lgB - a))) + j;
  xyzz_gstore(buf, idx, xyzz_add_with<FeAsm>(xyzz_gload(buf, idx), xyzz_gload(buf, idx + ((size_t)1 << lgh))));
}
__global__ __launch_bounds__(128) void k_halve_step(u32* __restrict__ buckets, int lgB, int t) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t id = tid >> 2;                     // one quad per addition
  const size_t total = (size_t)(t + 1) << (lgB - t - 1);
  if (id >= total) return;                        // quad-uniform
  halve_op(buckets + ((size_t)blockIdx.y << lgB) * 32, lgB, t, id, (int)(tid & 3));
}
// SEVERAL halving steps per launch.  Steps t0 .. t0 + s - 1 never pair two indices that differ by less than 2^(lgB - t0 - s), so
// every live region (region 0 = [0, W) and the regions a = 1 .. t0 at 2^(lgB - a), W = 2^(lgB - t0)) falls apart into
// 2^(lgB - t0 - s) independent problems of 2^s elements with that stride: the same recursion on a local array (region 0: every
// step leaves its upper half behind as a new block; the others only fold).  One workgroup loads its 2^s records (<= 32 KiB)
// into LDS, runs the s steps between barriers (DPP-quad additions, 128 quads) and stores back the s + 1 (or one) records
// that are still live -- instead of s dependent launches at ~6-10 us each whatever they hold (the 2^16 buckets of the
// 17-bit commit: thirteen launches, 91 us; now two launches of eight steps each).
constexpr int HMULTI_THREADS = 512;
constexpr int HMULTI_QUADS = HMULTI_THREADS / 4;
constexpr int HMULTI_MAX_S = 8;
__global__ __launch_bounds__(HMULTI_THREADS) void k_halve_multi(u32* __restrict__ buckets, int lgB, int t0, int s) {
  extern __shared__ __attribute__((aligned(16))) u32 lds_hm[];
  const int lgs = lgB - t0 - s;                    // log2 of the stride
  const int a = (int)(blockIdx.x >> lgs);          // region
  const size_t j0 = blockIdx.x & (((size_t)1 << lgs) - 1);
  u32* buf = buckets + ((size_t)blockIdx.y << lgB) * 32;
  const size_t base = ((a == 0) ? 0 : ((size_t)1 << (lgB - a))) + j0;
  const int ql = threadIdx.x & 3, quad = threadIdx.x >> 2;
  uint4* l4 = reinterpret_cast<uint4*>(lds_hm);
  uint4* g4 = reinterpret_cast<uint4*>(buf);
  for (int i = threadIdx.x; i < (8 << s); i += HMULTI_THREADS) l4[i] = g4[(base + ((size_t)(i >> 3) << lgs)) * 8 + (i & 7)];
  __syncthreads();
  for (int tl = 0; tl < s; tl++) {
    const int lgh = s - tl - 1;
    const int total = ((a == 0) ? (tl + 1) : 1) << lgh;
    for (int id = quad; id < total; id += HMULTI_QUADS) {        // quad-uniform
      const int aa = id >> lgh, j = id & ((1 << lgh) - 1);
      const int lo = ((aa == 0) ? 0 : (1 << (s - aa))) + j;
      const Xyzz x = xyzz_gload_quad(lds_hm, (size_t)lo, ql), y = xyzz_gload_quad(lds_hm, (size_t)(lo + (1 << lgh)), ql);
      xyzz_gstore_quad(lds_hm, (size_t)lo, xyzz_add_quad(x, y, ql), ql);
    }
    __syncthreads();
  }
  const int nlive = (a == 0) ? s + 1 : 1;          // local 0 and, in region 0, the heads 2^q of the blocks the steps left behind
  for (int i = threadIdx.x; i < 8 * nlive; i += HMULTI_THREADS) {
    const int e = i >> 3, k = (e == 0) ? 0 : (1 << (e - 1));
    g4[(base + ((size_t)k << lgs)) * 8 + (i & 7)] = l4[k * 8 + (i & 7)];
  }
}
#ifdef MZK_TUNING      // the DPP-quad tail of round 2: reachable only through MZK_ROW_TAILS=0 (A/B timing), not in the shipped library
constexpr int TAIL_THREADS = 512;
constexpr int TAIL_QUADS = TAIL_THREADS / 4;
// Remaining steps t_start..lgB-1 inside one workgroup per bucket set, then the weighted sum
// buf[0] + sum_j 2^j buf[2^j] (quad j doubles j times, LDS tree sum).  out[w] = XYZZ result of set w.
// finish_affine (single bucket set only): convert the sum to the canonical affine point here (one safegcd inversion on
// lane 0) instead of handing a 128-byte record to k_window_combine -- one dependent launch less on the merged path.
__global__ __launch_bounds__(TAIL_THREADS) void k_reduce_tail(u32* __restrict__ buckets, int lgB, int t_start, u32* __restrict__ out, int finish_affine) {
  __shared__ __attribute__((aligned(16))) u32 sh[32 * 32];
  u32* buf = buckets + ((size_t)blockIdx.x << lgB) * 32;
  const int lane = threadIdx.x & 3, quad = threadIdx.x >> 2;
  for (int t = t_start; t < lgB; t++) {
    const size_t total = (size_t)(t + 1) << (lgB - t - 1);
    for (size_t id = quad; id < total; id += TAIL_QUADS) halve_op(buf, lgB, t, id, lane);
    __syncthreads();
  }
  if (quad < 32) {
    Xyzz v = xyzz_inf();
    if (quad < lgB) {
      v = xyzz_gload_quad(buf, (size_t)1 << quad, lane);
      for (int d = 0; d < quad; d++) v = xyzz_dbl_quad(v, lane);
    } else if (quad == lgB) {
      v = xyzz_gload_quad(buf, 0, lane);
    }
    xyzz_gstore_quad(sh, quad, v, lane);
  }
  __syncthreads();
  for (int off = 16; off >= 1; off >>= 1) {
    if (quad < off) {
      const Xyzz a = xyzz_gload_quad(sh, quad, lane), b = xyzz_gload_quad(sh, quad + off, lane);
      xyzz_gstore_quad(sh, quad, xyzz_add_quad(a, b, lane), lane);
    }
    __syncthreads();
  }
  if (finish_affine) {
    if (threadIdx.x == 0) {
      u32 wds[16];
      Affine af;
      if (xyzz_to_affine<true>(xyzz_load(sh), &af)) affine_store_plain(af, wds);
      else for (int i = 0; i < 16; i++) wds[i] = 0;
      for (int i = 0; i < 16; i++) out[i] = wds[i];
    }
    return;
  }
  if (threadIdx.x < 32) out[(size_t)blockIdx.x * 32 + threadIdx.x] = sh[threadIdx.x];
}
#endif

// ---- small inputs (n < 4096): three launches instead of twenty-five ---------------------------------------------
// The reference's real callers commit to polynomials of at most a few thousand coefficients (das/avail.rs:96,
// das/eigenda.rs:99, every test in kzg.rs), where the general pipeline is nothing but launch latency: 25+ dependent
// launches of kernels that each run for microseconds.  Here:
//   k_small_sort        ONE workgroup: digits -> LDS histogram -> scan -> LDS cursors -> sorted entries (global)
//   k_small_accumulate  one WAVE per bucket: lanes stride over the bucket's entries (madd), then the 64 partials are
//                       summed by a quad-cooperative tree through LDS (seven rounds of ~2 us instead of a serial chain)
//   k_reduce_tail       all halving steps of a bucket set inside one workgroup (t_start = 0), then k_window_combine.
constexpr int SMALL_SORT_THREADS = 1024;
constexpr size_t SMALL_MAX_N = 4097;          // exclusive: 4096 (a blob of das/avail.rs) still takes the three-launch path
constexpr size_t SMALL_MAX_BUCKETS = 8192;
__global__ __launch_bounds__(SMALL_SORT_THREADS) void k_small_sort(const u32* __restrict__ scalars, size_t n, DigitLayout L, int NB,
                                                                    u32* __restrict__ offsets, u32* __restrict__ entries) {
  extern __shared__ u32 sh_small[];
  u32* hist = sh_small;                  // [NB]   counts, then cursors
  u32* scan = sh_small + NB;             // [SMALL_SORT_THREADS]
  const int tid = threadIdx.x;
  for (int b = tid; b < NB; b += SMALL_SORT_THREADS) hist[b] = 0;
  __syncthreads();
  for (size_t i = tid; i < n; i += SMALL_SORT_THREADS) {
    u32 w[8];
    load_scalar_canonical(scalars, i, w);
    walk_digits(w, L, i, [&](int, u32 key, u32) { atomicAdd(&hist[key], 1u); });
  }
  __syncthreads();
  const int per = (NB + SMALL_SORT_THREADS - 1) / SMALL_SORT_THREADS;
  u32 local = 0;
  for (int q = 0; q < per; q++) { const int b = tid * per + q; if (b < NB) local += hist[b]; }
  u32 all_entries;
  u32 run = block_excl_scan<SMALL_SORT_THREADS>(local, scan, &all_entries);
  for (int q = 0; q < per; q++) {
    const int b = tid * per + q;
    if (b < NB) { const u32 c = hist[b]; hist[b] = run; offsets[b] = run; run += c; }
  }
  if (tid == SMALL_SORT_THREADS - 1) offsets[NB] = all_entries;
  __syncthreads();
  for (size_t i = tid; i < n; i += SMALL_SORT_THREADS) {
    u32 w[8];
    load_scalar_canonical(scalars, i, w);
    walk_digits(w, L, i, [&](int, u32 key, u32 payload) { entries[atomicAdd(&hist[key], 1u)] = payload; });
  }
}
// The T per-lane partials of a bucket (the first min(cnt, T) can be non-trivial) -> their sum in buckets[b]: a tree in place
// through LDS, DPP quads while a level has more pairs than waves, one row-cooperative addition per wave below.
template <int T>
__device__ __forceinline__ void small_tree_store(u32* sh, const Xyzz& acc, u32 cnt, size_t b, u32* __restrict__ buckets) {
  const int lane = threadIdx.x;
  if (cnt > T) cnt = T;
  if (cnt <= 1) {            // workgroup-uniform
    if (lane == 0) xyzz_gstore(buckets, b, acc);
    return;
  }
  u32 width = 2;
  while (width < cnt) width <<= 1;            // partials beyond `cnt` are infinity: sum the first `width` only
  if ((u32)lane < width) xyzz_gstore(sh, lane, acc);
  __syncthreads();
  const int quad = lane >> 2, ql = lane & 3;
  const rowop::Lane ln = rowop::lane_init();
  for (u32 m = width; m > 1; m >>= 1) {
    if (m / 2 <= (u32)(T / 64)) {                         // the last levels: one row-cooperative addition per wave (1.3 us, not 3.6)
      for (u32 i = (u32)(lane >> 6); i < m / 2; i += T / 64) {
        const rowop::Pt x = rowop::load(sh + i * 32, ln), y = rowop::load(sh + (i + m / 2) * 32, ln);
        rowop::store(sh + i * 32, rowop::add(x, y, ln), ln);
      }
    } else {
      for (u32 i = quad; i < m / 2; i += T / 4) {         // quad-uniform
        const Xyzz x = xyzz_gload_quad(sh, i, ql), y = xyzz_gload_quad(sh, i + m / 2, ql);
        xyzz_gstore_quad(sh, i, xyzz_add_quad(x, y, ql), ql);
      }
    }
    __syncthreads();
  }
  if (lane < 32) buckets[b * 32 + lane] = sh[lane];
}

// T lanes per bucket (64: few entries per bucket -- the generic layout's thousands of buckets; 256: the commit against narrow
// window tables, ~256 entries per bucket at 2^10 coefficients: one entry per lane, so the per-lane chain of mixed additions
// (4 x ~4.5 us with 64 lanes) disappears and only the tree remains).  The tree adds partial i + m/2 onto partial i in place.
template <int T>
__global__ __launch_bounds__(T) void k_small_accumulate(const u32* __restrict__ points_mont, const u32* __restrict__ offsets,
                                                         const u32* __restrict__ entries, u32* __restrict__ buckets) {
  __shared__ __attribute__((aligned(16))) u32 sh[T * 32];
  const size_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const u32 o0 = offsets[b], o1 = offsets[b + 1];
  Xyzz acc = xyzz_inf();
  for (u32 e = o0 + lane; e < o1; e += T) {
    const u32 ent = entries[e];
    u32 w[16];
    const size_t idx = ent & 0x7fffffffu;
    load_words8(points_mont + idx * 16, w);
    load_words8(points_mont + idx * 16 + 8, w + 8);
    if (affine_words_is_inf(w)) continue;
    acc = xyzz_madd_signed_with<FeCpp>(acc, affine_load_mont(w), (ent >> 31) != 0);
  }
  small_tree_store<T>(sh, acc, o1 - o0, b, buckets);
}
// The same for a commit against window tables WITHOUT the sort launch: every workgroup walks all n scalars itself (n / T
// scalars per lane, ~8 instructions per digit: a few microseconds at n <= 4096 against the ~30 of k_small_sort's launch) and
// collects the entries of ITS bucket in an LDS list; entries beyond the list's capacity (skewed inputs: bit vectors, all
// scalars equal) are added on the spot by the lane that found them.  Merged layout only: the generic one would repeat the GLV
// split of every scalar in every workgroup.  The redundant walks grow with n * buckets: measured against the general pipeline
// (profiles/r03v_scan_path_size_sweep.txt) this path wins up to 2^14 coefficients at 10-bit windows (0.24 vs 0.38 ms at 2^13,
// 0.32 vs 0.39 at 2^14) and loses from 2^15 on (0.93 vs 0.39).
constexpr int SMALL_LIST_CAP = 4096;
template <int T, int C>
__global__ __launch_bounds__(T) void k_small_accumulate_scan(const u32* __restrict__ scalars, size_t n, size_t table_stride, const u32* __restrict__ points_mont,
                                                              u32* __restrict__ buckets) {
  constexpr int NWIN = 254 / C + 1;
  constexpr u32 HALF = 1u << (C - 1), MASKC = (1u << C) - 1u;
  __shared__ __attribute__((aligned(16))) u32 sh[T * 32];
  __shared__ u32 list[SMALL_LIST_CAP];
  __shared__ u32 count;
  const size_t b = blockIdx.x;
  const int lane = threadIdx.x;
  if (lane == 0) count = 0;
  __syncthreads();
  Xyzz acc = xyzz_inf();
  auto add_entry = [&](u32 ent) {
    u32 w[16];
    const size_t idx = ent & 0x7fffffffu;
    load_words8(points_mont + idx * 16, w);
    load_words8(points_mont + idx * 16 + 8, w + 8);
    if (affine_words_is_inf(w)) return;
    acc = xyzz_madd_signed_with<FeCpp>(acc, affine_load_mont(w), (ent >> 31) != 0);
  };
  // window values of this bucket: digit +(b + 1) and -(b + 1)
  const u32 vpos = (u32)b + HALF, vneg = HALF - 2u - (u32)b;
  for (size_t i = lane; i < n; i += T) {
    u32 w[8], t[9];
    load_scalar_canonical(scalars, i, w);
    u64 cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { cy += (u64)w[k] + digit_bias_word(C, k); t[k] = (u32)cy; cy >>= 32; }
    t[8] = (u32)cy + digit_bias_word(C, 8);
    u32 hits = 0;          // bit w: window w belongs to this bucket (NWIN <= 32)
    u32 negs = 0;
#pragma unroll
    for (int win = 0; win < NWIN; win++) {
      const int bit = win * C, k = bit >> 5, sft = bit & 31;
      const u64 pair = (u64)t[k] | ((k + 1 < 9) ? ((u64)t[k + 1] << 32) : 0ull);
      const u32 v = (u32)(pair >> sft) & MASKC;
      hits |= (v == vpos || v == vneg) ? (1u << win) : 0u;
      negs |= (v == vneg) ? (1u << win) : 0u;
    }
    while (hits) {
      const int win = __builtin_ctz(hits);
      hits &= hits - 1;
      const u32 payload = (u32)((size_t)win * table_stride + i) | (((negs >> win) & 1u) << 31);
      const u32 pos = atomicAdd(&count, 1u);
      if (pos < (u32)SMALL_LIST_CAP) list[pos] = payload;
      else add_entry(payload);
    }
  }
  __syncthreads();
  const u32 total = count;
  const u32 listed = total < (u32)SMALL_LIST_CAP ? total : (u32)SMALL_LIST_CAP;
  for (u32 e = lane; e < listed; e += T) add_entry(list[e]);
  small_tree_store<T>(sh, acc, total > (u32)SMALL_LIST_CAP ? (u32)T : total, b, buckets);
}
static_assert(254 / 8 + 1 <= 32, "window hit masks are 32 bits");

#ifdef MZK_TUNING
// k_window_combine / k_fold_partials (DPP-quad forms, round 2) live in mzk_msm_tail.hip: tuning build only
int launch_window_combine(const u32* wsum, int nwin, int c, int out_xyzz, u32* out, hipStream_t s);
int launch_fold_partials(const u32* partials, int count, u32* out, hipStream_t s);
#endif
// mzk_msm_row.hip: the same tails on row-cooperative group operations (one point operation per wave)
int launch_reduce_tail_row(u32* buckets, int lgB, int t_start, int sets, u32* out, int finish_affine, hipStream_t s);
int launch_window_combine_row(const u32* wsum, int nwin, int c, int out_xyzz, u32* out, hipStream_t s);
int launch_fold_partials_row(const u32* partials, int count, u32* out, hipStream_t s);
// tuning build: MZK_ROW_TAILS=0 selects the DPP-quad tails of round 2 (A/B timing: tools/timing/small_latency.py, time_msm.py)
#ifdef MZK_TUNING
static bool row_tails() {
  static const int v = tune_int("MZK_ROW_TAILS", 1);
  return v != 0;
}
#else
static constexpr bool row_tails() { return true; }
#endif
// The single-workgroup tail takes over once a halving step is at most this wide: a dependent launch costs ~6 us whatever runs in
// it (measured: a step of 1024 additions as one wave each 6.0 us, as DPP quads 6.4 us -- the launch, not the addition), a round
// of 256 quad additions inside the tail's workgroup 3.6 us, a round of 16 row additions ~1.8 us.
static size_t row_tail_max_ops() {
  static const size_t v = (size_t)tune_int("MZK_TAIL_MAX_OPS", 64);
  return v < 1 ? 1 : v;
}

// The halving steps that run as launches of their own, in front of the single-workgroup tail: one lane per addition while a step is
// throughput-bound (>= 2^16 additions over all sets), then k_halve_multi, up to eight steps per launch, until a step is at most
// `tail_max` additions wide.  *t_next = the first step left to the tail (lgB: all done, the tail only forms the weighted sum).
// MZK_HALVE_MULTI=0 (tuning build): one launch per step, the form of rounds 2-5.
static int launch_halving_steps(u32* buckets, int lgB, int sets, size_t tail_max, int* t_next, hipStream_t s) {
  static const int env_multi = tune_int("MZK_HALVE_MULTI", 1);
  int t = 0;
  while (t < lgB && ((size_t)(t + 1) << (lgB - t - 1)) > tail_max) {
    const size_t total = (size_t)(t + 1) << (lgB - t - 1);
    if (total * (size_t)sets >= ((size_t)1 << 16)) {
      hipLaunchKernelGGL(k_halve_step_wide, dim3((unsigned)((total + 127) / 128), (unsigned)sets), dim3(128), 0, s, buckets, lgB, t);
      t++;
      continue;
    }
    // several steps per launch only in the latency regime -- at most one workgroup per CU.  Its workgroups are unequal (region 0 carries
    // t + 1 times the additions of the others) and stay where they were placed, so with more of them than CUs the launches of ONE step,
    // which spread every step's additions over the whole chip, are faster (generic 2^20, 384 workgroups: bucket reduction 0.204 -> 0.222 ms;
    // 256 x 2^10 at 11 bits 0.196 -> 0.270; profiles/round6_halving_multi_and_rec4_ab.txt)
    const int st = (lgB - t < HMULTI_MAX_S) ? lgB - t : HMULTI_MAX_S;
    const size_t multi_wgs = ((size_t)(t + 1) << (lgB - t - st)) * (size_t)sets;
    if (env_multi == 0 || multi_wgs > (size_t)ctx().num_cu) {
      hipLaunchKernelGGL(k_halve_step, dim3((unsigned)((4 * total + 127) / 128), (unsigned)sets), dim3(128), 0, s, buckets, lgB, t);
      t++;
    } else {
      hipLaunchKernelGGL(k_halve_multi, dim3((unsigned)((size_t)(t + 1) << (lgB - t - st)), (unsigned)sets), dim3(HMULTI_THREADS), ((size_t)128 << st), s, buckets, lgB, t, st);
      t += st;
    }
  }
  MZK_HIP(hipGetLastError());
  *t_next = t;
  return MZK_OK;
}

// Bucket reduction of `sets` bucket sets of 2^lgB buckets each: sum_b (b+1) B_b by in-place halving -- wide steps as launches,
// the late ones inside one workgroup per set -- then, for the generic layout, the Horner over the windows.  merged: one set, the
// tail writes the result itself (affine point, or the XYZZ partial record).
static int reduce_bucket_sets(u32* buckets, int lgB, int sets, bool merged, int horner_c, u32* wsum, u32* d_out, bool out_partial_xyzz, hipStream_t s) {
  prof_begin(s, MZK_PH_MSM_REDUCE);
  const bool rows = row_tails();
  (void)rows;
  int t_start = 0;
#ifdef MZK_TUNING
  const size_t tail_max = rows ? row_tail_max_ops() : (size_t)4 * TAIL_QUADS;
#else
  const size_t tail_max = row_tail_max_ops();
#endif
  MZK_TRY(launch_halving_steps(buckets, lgB, sets, tail_max, &t_start, s));
  if (merged) {
#ifdef MZK_TUNING
    if (!rows) hipLaunchKernelGGL(k_reduce_tail, dim3(1), dim3(TAIL_THREADS), 0, s, buckets, lgB, t_start, d_out, out_partial_xyzz ? 0 : 1);
    else
#endif
    MZK_TRY(launch_reduce_tail_row(buckets, lgB, t_start, 1, d_out, out_partial_xyzz ? 0 : 1, s));
    MZK_HIP(hipGetLastError());
    prof_end(s, MZK_PH_MSM_REDUCE);
    return MZK_OK;
  }
#ifdef MZK_TUNING
  if (!rows) hipLaunchKernelGGL(k_reduce_tail, dim3((unsigned)sets), dim3(TAIL_THREADS), 0, s, buckets, lgB, t_start, wsum, 0);
  else
#endif
  MZK_TRY(launch_reduce_tail_row(buckets, lgB, t_start, sets, wsum, 0, s));
  MZK_HIP(hipGetLastError());
  prof_end(s, MZK_PH_MSM_REDUCE);
  prof_begin(s, MZK_PH_MSM_COMBINE);
#ifdef MZK_TUNING
  if (!rows) MZK_TRY(launch_window_combine((const u32*)wsum, sets, horner_c, out_partial_xyzz ? 1 : 0, d_out, s));
  else
#endif
  MZK_TRY(launch_window_combine_row((const u32*)wsum, sets, horner_c, out_partial_xyzz ? 1 : 0, d_out, s));
  prof_end(s, MZK_PH_MSM_COMBINE);
  return MZK_OK;
}

// ---- host orchestration -----------------------------------------------------------------------------------
// d_phi (optional): receives the endomorphism images (beta x, y) of the n points (the generic MSM layout reads them
// at index phi_offset + i)
int msm_prepare_points(const void* d_points_plain, size_t n, void* d_points_mont, void* d_phi, hipStream_t s) {
  if (n == 0) return MZK_OK;
  hipLaunchKernelGGL(k_prepare_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const u32*)d_points_plain,
                     (u32*)d_points_mont, n, (u32*)d_phi);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int msm_points_to_plain(const void* d_points_mont, size_t n, void* d_points_plain, hipStream_t s) {
  if (n == 0) return MZK_OK;
  hipLaunchKernelGGL(k_points_to_plain, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const u32*)d_points_mont, n, (u32*)d_points_plain);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int msm_phi_points(const void* d_points_mont, size_t n, void* d_phi, hipStream_t s) {
  if (n == 0) return MZK_OK;
  hipLaunchKernelGGL(k_phi_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const u32*)d_points_mont, n, (u32*)d_phi);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// SRS window tables: T[w][i] = 2^(c w) * P_i, affine Montgomery, w < nwin (w = 0 is the point itself).
// A per-point XYZZ state is doubled c times per window (k_srs_window_step) and converted to affine with the
// batched inversion of mzk_kzg.hip (one inversion per 16 points instead of one per table entry).
__global__ __launch_bounds__(128) void k_srs_state_init(const u32* __restrict__ pts_mont, size_t n, u32* __restrict__ state) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[16];
  load_words8(pts_mont + i * 16, w);
  load_words8(pts_mont + i * 16 + 8, w + 8);
  Xyzz p = affine_words_is_inf(w) ? xyzz_inf() : xyzz_from_affine(affine_load_mont(w));
  xyzz_gstore(state, i, p);
}
__global__ __launch_bounds__(128) void k_srs_window_step(u32* __restrict__ state, size_t n, int c) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Xyzz p = xyzz_gload(state, i);
  for (int d = 0; d < c; d++) p = xyzz_dbl_with<FeAsm>(p);
  xyzz_gstore(state, i, p);
}
int msm_build_tables(const void* d_points_mont, size_t n, void* d_tables, int window_bits, hipStream_t s) {
  if (n == 0) return MZK_OK;
  u32* state;
  MZK_TRY(ws_get(WS_XYZZ_TMP, n * 128, (void**)&state));
  const unsigned blocks = (unsigned)((n + 127) / 128);
  MZK_HIP(hipMemcpyAsync(d_tables, d_points_mont, n * 64, hipMemcpyDeviceToDevice, s));   // window 0
  hipLaunchKernelGGL(k_srs_state_init, dim3(blocks), dim3(128), 0, s, (const u32*)d_points_mont, n, state);
  for (int win = 1; win < msm_table_windows(window_bits); win++) {
    hipLaunchKernelGGL(k_srs_window_step, dim3(blocks), dim3(128), 0, s, state, n, window_bits);
    MZK_TRY(xyzz_batch_to_affine(state, n, (u32*)d_tables + (size_t)win * n * 16, true, s));
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// record-format dependent half of the two-level sort (coarse scatter, fine count, scan, fine scatter)
struct SortArgs {
  const u32* scalars; size_t n; DigitLayout L; int key_shift; u32 fine_mask; int fb; u32* binhist; int nwg; void* tmp; int F; int S;
  u32* finehist; u32* scan3; size_t sb_f; size_t n_fine; u32* offsets; u32* entries; size_t NBtot;
  unsigned fine_wgs; u32 fine_cap;       // grid of the fine kernels (slices of all bins + the extra slices of over-full ones) and the slice capacity (fine_plan)
  int cl;           // log2 of the coarse bins: COARSE_LOG, or 10 for the 20-bit merged layout
  // scan-free forms (null: the global scans of rounds 3-5): per-bin totals / cursors (in the call's cleared header), the bins + 1 bin starts the
  // coarse scatter publishes, per-bucket totals / cursors (zeroed by k_coarse_count)
  u32 *bin_tot, *bin_cur, *bin_start, *bucket_tot, *bucket_cur;
};
template <class REC>
static int sort_records(const SortArgs& a, hipStream_t s) {
  typedef typename REC::T R;
  const int cw = (a.L.merged && !a.L.glv && a.L.sets == 1 && (a.L.c == 16 || a.L.c == 17 || a.L.c == 20)) ? a.L.c : 0;     // the default widths by SRS size
  static const int env_staged = tune_int("MZK_COARSE_STAGED", 1);      // 0: A/B against the direct stores
  const unsigned bins = 1u << a.cl;
  // scan-free forms: a.bin_tot / a.bucket_tot non-null (see k_coarse_count, k_fine_scatter)
#define MZK_STAGED(C, CLOG) hipLaunchKernelGGL((k_coarse_scatter_staged<REC, C, CLOG>), dim3(a.nwg), dim3(SORT2_THREADS), lds, s, a.scalars, a.n, a.L.table_stride, \
                                               a.key_shift, a.fine_mask, a.fb, (const u32*)a.binhist, a.nwg, (R*)a.tmp, (const u32*)a.bin_tot, a.bin_cur, a.bin_start)
#define MZK_DIRECT(C) hipLaunchKernelGGL((k_coarse_scatter<REC, C>), dim3(a.nwg), dim3(SORT2_THREADS), 0, s, a.scalars, a.n, a.L, a.key_shift, a.fine_mask, a.fb, \
                                         (const u32*)a.binhist, a.nwg, (R*)a.tmp, (const u32*)a.bin_tot, a.bin_cur, a.bin_start)
  if (cw != 0 && (env_staged != 0 || a.cl != COARSE_LOG)) {
    const size_t lds = stage_lds_bytes(cw, a.cl, sizeof(R));
    bool& attr = ctx().attr_done[sizeof(R) == 4 ? ATTR_COARSE_STAGED4 : ATTR_COARSE_STAGED8];
    if (!attr) {
      MZK_HIP(hipFuncSetAttribute((const void*)k_coarse_scatter_staged<REC, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MZK_HIP(hipFuncSetAttribute((const void*)k_coarse_scatter_staged<REC, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MZK_HIP(hipFuncSetAttribute((const void*)k_coarse_scatter_staged<REC, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MZK_HIP(hipFuncSetAttribute((const void*)k_coarse_scatter_staged<REC, 20, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MZK_HIP(hipFuncSetAttribute((const void*)k_coarse_scatter_staged<REC, 17, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    if (cw == 17 && a.cl == 9) MZK_STAGED(17, 9);
    else if (cw == 20 && a.cl == 10) MZK_STAGED(20, 10);
    else if (cw == 20) MZK_STAGED(20, COARSE_LOG);
    else if (cw == 17) MZK_STAGED(17, COARSE_LOG);
    else MZK_STAGED(16, COARSE_LOG);
  }
#ifdef MZK_TUNING
  else if (cw == 20) MZK_DIRECT(20);
  else if (cw == 17) MZK_DIRECT(17);
  else if (cw == 16) MZK_DIRECT(16);
#endif
  else if (a.cl == 10) hipLaunchKernelGGL((k_coarse_scatter<REC, 0, 10>), dim3(a.nwg), dim3(SORT2_THREADS), 0, s, a.scalars, a.n, a.L, a.key_shift, a.fine_mask, a.fb,
                                          (const u32*)a.binhist, a.nwg, (R*)a.tmp, (const u32*)a.bin_tot, a.bin_cur, a.bin_start);
  else MZK_DIRECT(0);
#undef MZK_STAGED
#undef MZK_DIRECT
  // bin boundaries as the fine kernels index them: the coarse scan's [bin][workgroup] prefix, or the scan-free form's bins + 1 starts
  const u32* bounds = a.bin_tot ? (const u32*)a.bin_start : (const u32*)a.binhist;
  const int bstride = a.bin_tot ? 1 : a.nwg;
  const bool fine_free = a.bucket_tot != nullptr && a.F <= STAGE_F_MAX;
  hipLaunchKernelGGL((k_fine_count<REC>), dim3(a.fine_wgs), dim3(SORT2_THREADS), 0, s, (const R*)a.tmp, bounds, bstride, a.F, a.S,
                     a.fb, a.finehist, (int)bins, a.fine_cap, fine_free ? a.bucket_tot : (u32*)nullptr);
  if (!fine_free) MZK_TRY(launch_exclusive_scan((const u32*)a.finehist, a.finehist, a.n_fine, a.scan3, s));
  if (a.F <= STAGE_F_MAX) {
    const size_t lds = ((size_t)3 * a.F + SORT2_THREADS + STAGE_CAP) * 4 + (size_t)STAGE_CAP * 2;
    bool& staged_attr = ctx().attr_done[sizeof(R) == 4 ? ATTR_FINE_SCATTER4 : ATTR_FINE_SCATTER8];     // per instantiation and context
    if (!staged_attr) {
      MZK_HIP(hipFuncSetAttribute((const void*)k_fine_scatter<REC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      staged_attr = true;
    }
    hipLaunchKernelGGL((k_fine_scatter<REC>), dim3(a.fine_wgs), dim3(SORT2_THREADS), lds, s, (const R*)a.tmp, bounds, bstride, a.F,
                       a.S, a.fb, (const u32*)a.finehist, a.offsets, a.entries, a.NBtot, (int)bins, a.fine_cap,
                       fine_free ? (const u32*)a.bucket_tot : (const u32*)nullptr, fine_free ? a.bucket_cur : (u32*)nullptr);
  } else {
    hipLaunchKernelGGL((k_fine_scatter_direct<REC>), dim3(a.fine_wgs), dim3(SORT2_THREADS), 0, s, (const R*)a.tmp, bounds, bstride,
                       a.F, a.S, a.fb, (const u32*)a.finehist, a.offsets, a.entries, a.NBtot, (int)bins, a.fine_cap);
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// point_kind: 0 = affine canonical (ABI form), 1 = affine Montgomery (prepared), 2 = SRS window tables
// (SRS_WINDOWS x table_stride affine Montgomery points: all windows share one bucket set, no Horner).
// points_ready (optional, plain points only): called once, after the digit sort has been enqueued and before the first kernel
// that reads the points -- the host-buffer entry point stages the points onto the device there, so that the transfer of the
// 64 n bytes of points runs under the sort of the scalars instead of in front of it.
int msm_dev_impl(const void* d_scalars, const void* d_points, size_t n, int point_kind, size_t table_stride, void* d_out,
                 bool out_partial_xyzz, hipStream_t s, const std::function<int()>* points_ready, const MsmChunkCtx* cc) {
  if (!d_out || ((!d_scalars || !d_points) && n)) { set_error("msm: null pointer"); return MZK_E_ARG; }
  if (n > ((size_t)1 << 27)) { set_error("msm: n > 2^27 not supported"); return MZK_E_ARG; }
  // chunk mode (msm_chunked_impl): the pairs [i0, i0 + n) of a problem of n_shape pairs; layout by the whole problem, per-chunk buffers
  // sized for the largest chunk, returns with the chunk's buckets summed (no reduction)
  const size_t n_shape = cc ? cc->n_total : n, n_alloc = cc ? cc->n_alloc : n, i0 = cc ? cc->i0 : 0;
  const int ck = cc ? cc->k : 0, cK = cc ? cc->K : 1;
  hipStream_t ss = (cc && cc->sort_stream) ? cc->sort_stream : s;      // the stream of the digit sort
  if (cc && (n < SMALL_MAX_N || n > n_alloc)) { set_error("msm: chunk of %zu pairs (chunks hold 4097 .. %zu)", n, n_alloc); return MZK_E_ARG; }
  if (n == 0) {  // empty polynomial -> point at infinity (polynomial.rs:160)
    if (points_ready) MZK_TRY((*points_ready)());
    MZK_HIP(hipMemsetAsync(d_out, 0, out_partial_xyzz ? 128 : 64, s));
    return MZK_OK;
  }
  DigitLayout L;
  const int table_c = ((point_kind >> 8) & 0xff) ? ((point_kind >> 8) & 0xff) : 16;
  const int table_sets = ((point_kind >> 16) & 0xff) ? ((point_kind >> 16) & 0xff) : 1;
  point_kind &= 0xff;
  L.merged = (point_kind == 2) ? 1 : 0;
  L.sets = L.merged ? table_sets : 1;
  L.table_stride = table_stride;
  // generic layout: GLV split (mzk_glv.h) -- 2n points (P_i and phi(P_i) at phi_offset + i), half-length scalars
  L.glv = L.merged ? 0 : 1;
  L.phi_offset = (point_kind == 0) ? n_shape : table_stride;   // prepared below / laid out by the SRS handle
  MsmShape sh = L.merged ? choose_shape(n_shape) : choose_shape_glv(n_shape);
  if (L.merged) {
    sh.c = table_c; sh.nwin = msm_table_windows(table_c); sh.lgB = sh.c - 1; sh.nbuckets = (size_t)L.sets << sh.lgB;
  }
  L.c = sh.c; L.nwin = sh.nwin;
  const size_t NB = sh.nbuckets;
  const bool one_set = L.merged && L.sets == 1;     // the tail writes the result itself: no Horner over bucket sets
  const int red_windows = L.merged ? L.sets : sh.nwin;   // bucket sets to reduce
  const int horner_c = one_set ? 0 : sh.c;
  // (a chunk's entries count from its first pair: every table row / the endomorphism images sit at the same distance behind it)
  const u32* pts = (const u32*)d_points + i0 * 16;
  void* pm = nullptr;
  if (point_kind == 0) {
    MZK_TRY(ws_get(WS_MSM_POINTS, 2 * n_shape * 64, &pm));
    pts = (const u32*)pm + i0 * 16;
  }
  bool prepared = false;
  auto prepare = [&]() -> int {          // Montgomery form + endomorphism images of plain points; once, before their first reader
    if (prepared) return MZK_OK;
    prepared = true;
    if (points_ready) MZK_TRY((*points_ready)());
    if (point_kind != 0) return MZK_OK;
    prof_begin(s, MZK_PH_MSM_PREPARE);
    MZK_TRY(msm_prepare_points((const u32*)d_points + i0 * 16, n, (u32*)pm + i0 * 16, (u32*)pm + (n_shape + i0) * 16, s));
    prof_end(s, MZK_PH_MSM_PREPARE);
    return MZK_OK;
  };
  const size_t windows_per_pair = (size_t)(L.glv ? 2 * sh.nwin : sh.nwin);
  const size_t E_max = n * windows_per_pair, E_alloc = n_alloc * windows_per_pair;
  // (the generic layout has twice the entries per pair: measured at 4096 pairs it is 5 % slower on this path, the commit 14 % faster)
  static const int env_scan = tune_int("MZK_SMALL_SCAN", 1);     // 0: A/B against the sorted path
  static const int env_scan_log = tune_int("MZK_SCAN_MAX_LOG", 14);
  const bool scan_ok = one_set && env_scan != 0 && n <= ((size_t)1 << env_scan_log) && (L.c == 8 || (L.c >= 10 && L.c <= 13));
  if (!cc && (scan_ok || n < (L.merged ? SMALL_MAX_N : SMALL_MAX_N - 1)) && NB <= SMALL_MAX_BUCKETS) {
    u32 *offsets, *entries, *buckets, *wsum;
    MZK_TRY(ws_get(WS_MSM_OFFSETS, (NB + 1) * 4, (void**)&offsets));
    MZK_TRY(ws_get(WS_MSM_ENTRIES, E_max * 4, (void**)&entries));
    MZK_TRY(ws_get(WS_MSM_BUCKETS, NB * 128, (void**)&buckets));
    MZK_TRY(ws_get(WS_MSM_OUT, (size_t)MAX_WINDOWS * 128, (void**)&wsum));
    MZK_TRY(prepare());
    if (scan_ok) {      // two launches: every bucket's workgroup finds its own entries
      prof_begin(s, MZK_PH_MSM_ACCUMULATE);
#define MZK_SCAN_CASE(C) case C: hipLaunchKernelGGL((k_small_accumulate_scan<256, C>), dim3((unsigned)NB), dim3(256), 0, s, (const u32*)d_scalars, n, L.table_stride, pts, buckets); break;
      switch (L.c) { MZK_SCAN_CASE(8) MZK_SCAN_CASE(10) MZK_SCAN_CASE(11) MZK_SCAN_CASE(12) MZK_SCAN_CASE(13) }
#undef MZK_SCAN_CASE
      prof_end(s, MZK_PH_MSM_ACCUMULATE);
      MZK_TRY(reduce_bucket_sets(buckets, sh.lgB, red_windows, true, horner_c, wsum, (u32*)d_out, out_partial_xyzz, s));
      MZK_HIP(hipGetLastError());
      return MZK_OK;
    }
    prof_begin(s, MZK_PH_MSM_SORT);
    hipLaunchKernelGGL(k_small_sort, dim3(1), dim3(SMALL_SORT_THREADS), (NB + SMALL_SORT_THREADS) * 4, s, (const u32*)d_scalars, n, L, (int)NB, offsets, entries);
    prof_end(s, MZK_PH_MSM_SORT);
    prof_begin(s, MZK_PH_MSM_ACCUMULATE);
    if (E_max / NB > 64)      // ~256 entries per bucket (commits against narrow tables): a lane per entry
      hipLaunchKernelGGL(k_small_accumulate<256>, dim3((unsigned)NB), dim3(256), 0, s, pts, (const u32*)offsets, (const u32*)entries, buckets);
    else
      hipLaunchKernelGGL(k_small_accumulate<64>, dim3((unsigned)NB), dim3(64), 0, s, pts, (const u32*)offsets, (const u32*)entries, buckets);
    prof_end(s, MZK_PH_MSM_ACCUMULATE);
    MZK_TRY(reduce_bucket_sets(buckets, sh.lgB, red_windows, one_set, horner_c, wsum, (u32*)d_out, out_partial_xyzz, s));
    MZK_HIP(hipGetLastError());
    return MZK_OK;
  }
  // entries pack the point reference into 31 bits (+ sign) and entry positions into 32: reject shapes that overflow
  // (window widths below 16 on a > 2^26-point SRS) instead of gathering a wrong table row
  {
    const size_t ref_limit = L.merged ? (size_t)msm_table_rows(sh.c, L.sets) * table_stride : L.phi_offset + n;
    if (ref_limit > ((size_t)1 << 31) || E_max >= ((size_t)1 << 32)) {
      set_error("msm: %zu pairs x %d windows (table stride %zu) exceed the 31-bit point references / 32-bit entry offsets", n, sh.nwin, table_stride);
      return MZK_E_ARG;
    }
  }
  // Segment length: one lane per segment, sized as if four waves per SIMD were resident (E / (CUs * 4 * 4 * 64), >= 16).  Round 2's
  // kernel had 123 VGPRs and the grid was exactly one round of resident waves; with the signed mixed addition of round 3 the
  // compiler takes 140 VGPRs (three waves per SIMD), and the same segment length is still the fastest of the variants measured
  // (profiles/r03o_accumulate_occupancy_ab.txt: shipped 1.05-1.06 ms at 2^20; __launch_bounds__(256, 4) = 128 VGPRs + 64 B of
  // scratch 1.07-1.10; segments sized for three waves 1.07 with a cheaper segment combine: equal in total).
  // (tools/timing/acc_sweep.py sweeps MZK_ACC_PREFETCH / MZK_ACC_SEG.)
  static const int env_prefetch = tune_int("MZK_ACC_PREFETCH", -1);
  static const int env_seg = tune_int("MZK_ACC_SEG", 0);
  const bool acc_prefetch = env_prefetch >= 0 ? env_prefetch != 0 : false;
  const size_t resident_lanes = (size_t)ctx().num_cu * 4 * (acc_prefetch ? 3 : 4) * 64;
  size_t seg_sz = (E_max + resident_lanes - 1) / resident_lanes;
  if (seg_sz < 16) seg_sz = 16;
  if (env_seg > 0) seg_sz = (size_t)env_seg;
  const u32 seg = (u32)seg_sz;
  const size_t T = (E_max + seg_sz - 1) / seg_sz;
  const u32 t_max = env_seg > 0 ? 0u : (u32)T;            // the kernels shorten the segments when the scalars emit fewer entries (segment_length)
  const size_t nslots = T + NB + 1;
  const size_t max_heavy = (T + NB) / HEAVY_SLOTS + 1;             // at most (T + NB) / 33 buckets hold more than 32 partials
  const size_t heavy_words = heavy_total_words(max_heavy);
  u32 *counts, *offsets, *ranks, *entries, *scan_tmp, *buckets, *slots;
  // two-level sort when the bucket space is a power of two >= 2^12 (merged layout always; generic at c = 16)
  // The two-level sort wants a power of two: the generic layout's 7 x 2^18 buckets (19-bit windows) sort as if there were an eighth,
  // empty window -- the sort's arrays are sized by NBtot, its offsets beyond NB all equal the entry count, everything after the sort
  // works on the NB real buckets.
  size_t NBtot = NB;
  if (L.glv && (NB & (NB - 1)) != 0 && sh.c >= 17) { NBtot = 1; while (NBtot < NB) NBtot <<= 1; }
  MZK_TRY(ws_get(WS_MSM_COUNTS, 2 * NBtot * 4, (void**)&counts));      // (two-level sort, scan-free form: per-bucket totals + cursors)
  // per-chunk buffers (chunk mode: cK of each, one behind the other; the sort's scratch is shared -- the chunks' sorts run one after
  // the other on one stream, and nothing after a chunk's sort reads it)
  MZK_TRY(ws_get(WS_MSM_OFFSETS, (size_t)cK * (NBtot + 1) * 4, (void**)&offsets));
  offsets += (size_t)ck * (NBtot + 1);
  // (small inputs keep the one-pass kernels, except that the merged one-pass histogram must fit the LDS: 2^15 buckets)
  // coarse bins: 256, or 1024 for the 20-bit merged layout (2^19 buckets: 512 per bin instead of 2048; k_coarse_count)
  static const int env_cl20 = tune_int("MZK_COARSE_LOG_20", 10);      // tuning build: 8 = the 256-bin form at 20 bits too
  // (the generic GLV layout stays at 256 bins: its coarse scatter stores records one by one -- the walk is the GLV split, no staging --
  // and 512 bins measured slower at 2^22 and 2^24: sort 3.77 -> 3.91 ms, profiles/round5_sort_1024_bins.txt)
  int cl = (L.merged && !L.glv && L.sets == 1 && L.c == 20 && env_cl20 == 10) ? 10 : COARSE_LOG;
  // 17-bit merged layout: 512 bins when that one bit is what lets the sort's intermediate records shrink from 8 to 4 bytes (reference
  // 15 n < 2^24, 7-bit fine key, sign: 2^20 pairs exactly) -- the coarse scatter writes and both fine passes read half the bytes
  // (profiles/round6_halving_multi_and_rec4_ab.txt; without that gain 512 bins lost to 256 in round 3: HISTORY)
  static const int env_cl17 = tune_int("MZK_COARSE_LOG_17", 9);       // tuning build: 8 = the 256-bin form with 8-byte records
  if (L.merged && !L.glv && L.sets == 1 && L.c == 17 && env_cl17 == 9 && cl == COARSE_LOG && COARSE_LOG == 8) {
    const size_t ref_max17 = (size_t)msm_table_rows(sh.c, L.sets) * table_stride;
    if (ref_max17 > ((size_t)1 << (31 - 8)) && ref_max17 <= ((size_t)1 << (31 - 7))) cl = 9;
  }
  if (L.glv && NBtot > ((size_t)STAGE_F_MAX << COARSE_LOG)) cl = 10;      // generic layout at 19 bits: 2^21 sorted buckets, 2048 per bin
  const size_t cbins = (size_t)1 << cl;
  const bool two_level = (NBtot & (NBtot - 1)) == 0 && NBtot >= 4096 && (NBtot / cbins) <= (size_t)FINE_MAX &&
                         (n >= 4096 || ((point_kind & 0xff) == 2 && NBtot > ((size_t)1 << 15)));
  if (cc && !two_level) { set_error("msm: chunk mode needs the two-level sort (layout %d bits, %zu buckets)", sh.c, NB); return MZK_E_ARG; }
  MZK_TRY(ws_get(WS_MSM_CURSOR, E_alloc * (two_level ? 8 : 4), (void**)&ranks));
  MZK_TRY(ws_get(WS_MSM_ENTRIES, (size_t)cK * E_alloc * 4, (void**)&entries));
  entries += (size_t)ck * E_alloc;
  MZK_TRY(ws_get(WS_MSM_SCAN, scan_scratch_words(NB) * 4, (void**)&scan_tmp));
  MZK_TRY(ws_get(WS_MSM_BUCKETS, (size_t)cK * NB * 128, (void**)&buckets));
  buckets += (size_t)ck * NB * 32;
  // (a chunk never has more segments than resident lanes, nor than entries / 16: the slot region of every chunk is sized for that)
  const size_t T_cap = cc ? ((resident_lanes + 1 < E_alloc / 16 + 1) ? resident_lanes + 1 : E_alloc / 16 + 1) : T;
  const size_t nslots_cap = T_cap + NB + 1, heavy_words_cap = heavy_total_words((T_cap + NB) / HEAVY_SLOTS + 1);
  const size_t slot_region_words = (nslots_cap * SLOT_WORDS + heavy_words_cap + 3) & ~(size_t)3;
  MZK_TRY(ws_get(WS_MSM_SLOTS, (size_t)cK * slot_region_words * 4, (void**)&slots));
  slots += (size_t)ck * slot_region_words;
  u32* heavy = slots + nslots * SLOT_WORDS;
  // the digit sort runs on `ss`: behind everything the main stream has enqueued so far (the workspace's previous users), and the
  // accumulate on the main stream behind it
  if (ss != s) {
    hipEvent_t ev;
    MZK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const hipError_t e1 = hipEventRecord(ev, s), e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(ss, ev, 0) : e1;
    (void)hipEventDestroy(ev);
    MZK_HIP(e2);
  }
  hipStream_t s_main = s;
  s = ss;              // (the sort phase below is written against `s`)
  prof_begin(s, MZK_PH_MSM_SORT);
  // every slot k_seg_combine reads is written by k_seg_accumulate first (slot t + b exists exactly when segment t
  // overlaps bucket b; checked by poisoning the array under the whole GPU suite), so only the heavy-bucket counter
  // needs clearing
  MZK_HIP(hipMemsetAsync(heavy, 0, HEAVY_CLEAR_BYTES, s));
  const unsigned nblk = (unsigned)((n + 255) / 256);
  if (two_level) {
    int kb = 0;
    while (((size_t)1 << kb) < NBtot) kb++;
    const int key_shift = kb - cl;
    const int F = (int)(NBtot / cbins);
    const u32 fine_mask = (u32)F - 1u;
    const int nwg = (int)((n + COARSE_PER_WG - 1) / COARSE_PER_WG);
    // records per fine workgroup: 32 Ki for the merged layout, 16 Ki for the generic one (measured: generic sort 0.250 -> 0.230 ms
    // at 2^20, merged equal within noise from 16 Ki to 64 Ki: profiles/r04m_*), 128 Ki when a bin has thousands of buckets
    // (the [bucket][sub] histogram that is scanned afterwards has NB * S entries)
    static const int env_per_fine = tune_int("MZK_PER_FINE", 0);      // tuning: tools/timing/window_sweep.py
    const size_t per_fine = env_per_fine > 0 ? (size_t)env_per_fine : ((NBtot / cbins >= 4096) ? 131072 : (NBtot / cbins >= 2048) ? 65536 : (L.glv ? 16384 : 32768));
    int S = (int)((E_max / cbins + per_fine - 1) / per_fine);
    if (S < 2) S = 2;
    if (S > 64) S = 64;
    // fine workgroups: S slices for every bin + the extra slices of over-full bins (fine_plan: a slice holds at most 1.5 nominal ones)
    const size_t slice_nom = (E_max + cbins * (size_t)S - 1) / (cbins * (size_t)S);
    const size_t slice_cap = slice_nom + slice_nom / 2 + 1;
    const size_t fine_wgs = cbins * (size_t)S + (E_max + slice_cap - 1) / slice_cap + 1;
    const size_t n_coarse = cbins * nwg, n_fine = (size_t)F * fine_wgs;
    u32 *binhist, *finehist;
    // (chunk mode: sized for the largest chunk, the same request in every chunk's call)
    const size_t nwg_a = (n_alloc + COARSE_PER_WG - 1) / COARSE_PER_WG, n_coarse_a = cbins * nwg_a;
    const size_t slice_nom_a = (E_alloc + cbins * (size_t)64 - 1) / (cbins * (size_t)64);      // (S <= 64: the smallest nominal slice)
    const size_t fine_wgs_a = cc ? cbins * 64 + (E_alloc + slice_nom_a) / (slice_nom_a + slice_nom_a / 2 + 1) + 2 : fine_wgs;
    const size_t n_fine_a = cc ? (size_t)F * fine_wgs_a : n_fine;
    MZK_TRY(ws_get(WS_MSM_WGHIST, ((cc ? n_coarse_a : n_coarse) + 1 + n_fine_a + 1 + cbins + 1) * 4, (void**)&binhist));
    finehist = binhist + (cc ? n_coarse_a : n_coarse) + 1;
    const size_t sb_f = (n_fine + SCAN_BLOCK - 1) / SCAN_BLOCK;
    u32* scan2;
    MZK_TRY(ws_get(WS_MSM_SCAN, (scan_scratch_words(cc ? n_coarse_a : n_coarse) + scan_scratch_words(n_fine_a) + 4) * 4, (void**)&scan2));
    // scan-free sort (k_coarse_count, k_fine_scatter): no global scan at either level, nine launches -> five
    static const int env_scan_free = tune_int("MZK_SORT_SCAN_FREE", 3);       // tuning build: bit 0 = coarse level, bit 1 = fine level
    const bool coarse_free = (env_scan_free & 1) != 0 && cbins <= (size_t)SORT_CTR_BINS;
    const bool fine_free = (env_scan_free & 2) != 0 && F <= STAGE_F_MAX;
    u32* bin_tot = coarse_free ? heavy + SORT_CTR_AT : nullptr;
    u32* bin_cur = coarse_free ? heavy + SORT_CTR_AT + SORT_CTR_BINS : nullptr;
    u32* bin_start = coarse_free ? finehist + n_fine_a + 1 : nullptr;
    u32* bucket_tot = fine_free ? counts : nullptr;                          // (counts: 2 NB words, see above)
    u32* bucket_cur = fine_free ? counts + NBtot : nullptr;
    u32* zero_ptr = fine_free ? counts : nullptr;
    const size_t zero_words = fine_free ? 2 * NBtot : 0;
#define MZK_CCOUNT(C, CLOG) hipLaunchKernelGGL((k_coarse_count<C, CLOG>), dim3(nwg), dim3(SORT2_THREADS), 0, s, (const u32*)d_scalars, n, L, key_shift, binhist, nwg, \
                                               bin_tot, zero_ptr, zero_words)
    const bool plain_merged = L.merged && !L.glv && L.sets == 1;
    if (cl == 10 && L.glv) MZK_CCOUNT(0, 10);
    else if (cl == 10) MZK_CCOUNT(20, 10);
    else if (cl == 9) MZK_CCOUNT(17, 9);
    else if (plain_merged && L.c == 20) MZK_CCOUNT(20, COARSE_LOG);
    else if (plain_merged && L.c == 17) MZK_CCOUNT(17, COARSE_LOG);
    else if (plain_merged && L.c == 16) MZK_CCOUNT(16, COARSE_LOG);
    else MZK_CCOUNT(0, COARSE_LOG);
#undef MZK_CCOUNT
    if (!coarse_free) MZK_TRY(launch_exclusive_scan((const u32*)binhist, binhist, n_coarse, scan2, s));
    int fb = 0;
    while ((1 << fb) < F) fb++;
    // largest point reference: merged nwin * stride, generic phi_offset + n
    const size_t ref_max = L.merged ? (size_t)msm_table_rows(sh.c, L.sets) * table_stride : L.phi_offset + n;
    const bool compact = ref_max <= ((size_t)1 << (31 - fb));      // references are < ref_max
    SortArgs sa{(const u32*)d_scalars, n, L, key_shift, fine_mask, fb, binhist, nwg, (void*)ranks, F, S, finehist, scan2 + scan_scratch_words(cc ? n_coarse_a : n_coarse) + 2, sb_f, n_fine,
                offsets, entries, NBtot, (unsigned)fine_wgs, (u32)slice_cap, cl, bin_tot, bin_cur, bin_start, bucket_tot, bucket_cur};
    MZK_TRY(compact ? sort_records<Rec4>(sa, s) : sort_records<Rec8>(sa, s));
  } else if (L.merged) {
    // LDS histogram path (no global atomics)
    int nwg = (int)((n + 4095) / 4096);
    if (nwg > ctx().num_cu) nwg = ctx().num_cu;
    if (nwg < 1) nwg = 1;
    const size_t per_wg = (n + nwg - 1) / nwg;
    u32* wg_hist;
    MZK_TRY(ws_get(WS_MSM_WGHIST, (size_t)nwg * NB * 4, (void**)&wg_hist));
    const size_t lds = NB * 4;
    bool& lds_attr_set = ctx().attr_done[ATTR_DIGITS_LDS];
    if (!lds_attr_set) {
      MZK_HIP(hipFuncSetAttribute((const void*)k_digits_count_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MZK_HIP(hipFuncSetAttribute((const void*)k_digits_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      lds_attr_set = true;
    }
    hipLaunchKernelGGL(k_digits_count_lds, dim3(nwg), dim3(LDS_SORT_THREADS), lds, s, (const u32*)d_scalars, n, per_wg, L, wg_hist, ranks);
    hipLaunchKernelGGL(k_wg_hist_prefix, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, s, wg_hist, nwg, (int)NB, counts);
    MZK_TRY(launch_exclusive_scan((const u32*)counts, offsets, NB, scan_tmp, s));
    hipLaunchKernelGGL(k_digits_scatter_lds, dim3(nwg), dim3(LDS_SORT_THREADS), lds, s, (const u32*)d_scalars, n, per_wg, L, (const u32*)offsets,
                       (const u32*)wg_hist, (const u32*)ranks, entries);
  } else {
    MZK_HIP(hipMemsetAsync(counts, 0, NB * 4, s));
    hipLaunchKernelGGL(k_digits_count, dim3(nblk), dim3(256), 0, s, (const u32*)d_scalars, n, L, counts, ranks);
    MZK_TRY(launch_exclusive_scan((const u32*)counts, offsets, NB, scan_tmp, s));
    hipLaunchKernelGGL(k_digits_scatter, dim3(nblk), dim3(256), 0, s, (const u32*)d_scalars, n, L, offsets, ranks, entries);
  }
  MZK_HIP(hipGetLastError());
  prof_end(s, MZK_PH_MSM_SORT);
  if (s != s_main) {       // the accumulate (main stream) behind this chunk's sort
    hipEvent_t ev;
    MZK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const hipError_t e1 = hipEventRecord(ev, s), e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(s_main, ev, 0) : e1;
    (void)hipEventDestroy(ev);
    MZK_HIP(e2);
  }
  s = s_main;
  MZK_TRY(prepare());
  prof_begin(s, MZK_PH_MSM_ACCUMULATE);
  // the true entry count lives in offsets[NB] on the device; lanes past it exit (E_max bounds it)
#ifdef MZK_TUNING
  if (acc_prefetch)
    hipLaunchKernelGGL(k_seg_accumulate<true>, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, pts, offsets, entries, slots, NB, seg, t_max);
  else
#endif
    hipLaunchKernelGGL(k_seg_accumulate<false>, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, pts, offsets, entries, slots, NB, seg, t_max);
  prof_end(s, MZK_PH_MSM_ACCUMULATE);
  prof_begin(s, MZK_PH_MSM_SEG_COMBINE);
  static const int wide_min_log = tune_int("MZK_COMBINE_WIDE_MIN_LOG", 17);
  if (NB >= ((size_t)1 << wide_min_log))
    hipLaunchKernelGGL(k_seg_combine_wide, dim3((unsigned)((NB + 127) / 128)), dim3(128), 0, s, slots, offsets, buckets, NB, seg, heavy, t_max);
  else
    hipLaunchKernelGGL(k_seg_combine<1>, dim3((unsigned)((4 * NB + 127) / 128)), dim3(128), 0, s, slots, offsets, buckets, NB, seg, heavy, (const u32*)nullptr, 0, t_max);
  hipLaunchKernelGGL(k_seg_combine_heavy, dim3(HEAVY_GRID), dim3(HEAVY_THREADS), 0, s, (const u32*)slots, (const u32*)offsets, buckets, heavy, max_heavy);
  MZK_HIP(hipGetLastError());
  prof_end(s, MZK_PH_MSM_SEG_COMBINE);
  if (cc) return MZK_OK;        // chunk mode: msm_chunked_impl sums the chunks' bucket arrays and reduces once

  u32* wsum;
  MZK_TRY(ws_get(WS_MSM_OUT, (size_t)MAX_WINDOWS * 128, (void**)&wsum));
  MZK_TRY(reduce_bucket_sets(buckets, sh.lgB, red_windows, one_set, horner_c, wsum, (u32*)d_out, out_partial_xyzz, s));
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// ---- one MSM in chunks -------------------------------------------------------------------------------------------------------------
// buckets[b] += buckets[k NB + b], k = 1 .. K-1 (one lane per bucket)
__global__ __launch_bounds__(128) void k_fold_bucket_sets(u32* __restrict__ buckets, size_t NB, int K) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= NB) return;
  Xyzz acc = xyzz_gload(buckets, b);
  for (int k = 1; k < K; k++) acc = xyzz_add_with<FeAsm>(acc, xyzz_gload(buckets, (size_t)k * NB + b));
  xyzz_gstore(buckets, b, acc);
}
// Chunk mode covers the layouts of the large calls: window tables with one bucket set, and the generic GLV layout -- from 2^18 pairs on
// (every chunk must take the two-level sort and keep the accumulate's lanes busy).
bool msm_chunkable(size_t n_total, int point_kind) {
  const int kind = point_kind & 0xff, sets = ((point_kind >> 16) & 0xff) ? ((point_kind >> 16) & 0xff) : 1;
  return n_total >= ((size_t)1 << 18) && (kind != MSM_PTS_TABLES || sets == 1);
}
// Sum over the chunks' pairs = the MSM of all n_total pairs (polynomial.rs:156-165 is a sum over independent pairs; the bucket sums
// are too).  chunks[k].d_scalars: that chunk's scalars on the device; d_points: the WHOLE point array / table set.  A chunk is sorted
// (on sort_stream if given -- then chunk k + 1's sort may run under chunk k's accumulate) and accumulated as soon as its `ready`
// event has fired: the host-buffer entry points upload chunk k + 1 meanwhile.  Same canonical affine point as the one-piece call.
int msm_chunked_impl(const MsmChunk* chunks, int K, const void* d_points, size_t n_total, int point_kind, size_t table_stride, void* d_out,
                     bool out_partial_xyzz, hipStream_t s, hipStream_t sort_stream, const std::function<int(int)>* before_chunk) {
  if (!chunks || K < 1 || K > 8 || !d_out || !d_points) { set_error("msm_chunked: bad argument"); return MZK_E_ARG; }
  size_t n_alloc = 0, covered = 0;
  for (int k = 0; k < K; k++) {
    if (chunks[k].i0 != covered) { set_error("msm_chunked: chunks must be consecutive"); return MZK_E_ARG; }
    covered += chunks[k].n;
    n_alloc = chunks[k].n > n_alloc ? chunks[k].n : n_alloc;
  }
  if (covered != n_total || !msm_chunkable(n_total, point_kind)) { set_error("msm_chunked: chunks do not cover a chunkable problem"); return MZK_E_ARG; }
  for (int k = 0; k < K; k++) {
    // before_chunk(k): the caller brings chunk k's inputs onto the device (a host-buffer entry point copies them here, blocking the
    // host while the GPU works on chunk k - 1) and records chunks[k].ready
    if (before_chunk) MZK_TRY((*before_chunk)(k));
    if (chunks[k].ready) MZK_HIP(hipStreamWaitEvent(sort_stream ? sort_stream : s, chunks[k].ready, 0));
    if (chunks[k].ready && sort_stream && (point_kind & 0xff) == MSM_PTS_PLAIN) MZK_HIP(hipStreamWaitEvent(s, chunks[k].ready, 0));     // the points are read on the main stream
    const MsmChunkCtx cc{k, K, chunks[k].i0, n_total, n_alloc, sort_stream};
    MZK_TRY(msm_dev_impl(chunks[k].d_scalars, d_points, chunks[k].n, point_kind, table_stride, d_out, out_partial_xyzz, s, nullptr, &cc));
  }
  // the layout every chunk used (msm_dev_impl derives the same from n_total)
  const int kind = point_kind & 0xff;
  const int table_c = ((point_kind >> 8) & 0xff) ? ((point_kind >> 8) & 0xff) : 16;
  const bool merged = kind == MSM_PTS_TABLES;
  MsmShape sh = merged ? choose_shape(n_total) : choose_shape_glv(n_total);
  if (merged) { sh.c = table_c; sh.nwin = msm_table_windows(table_c); sh.lgB = sh.c - 1; sh.nbuckets = (size_t)1 << sh.lgB; }
  const size_t NB = sh.nbuckets;
  u32 *buckets, *wsum;
  MZK_TRY(ws_get(WS_MSM_BUCKETS, (size_t)K * NB * 128, (void**)&buckets));       // (the chunks' request: same size, same buffer)
  MZK_TRY(ws_get(WS_MSM_OUT, (size_t)MAX_WINDOWS * 128, (void**)&wsum));
  if (K > 1) {
    prof_begin(s, MZK_PH_MSM_SEG_COMBINE);
    hipLaunchKernelGGL(k_fold_bucket_sets, dim3((unsigned)((NB + 127) / 128)), dim3(128), 0, s, buckets, NB, K);
    MZK_HIP(hipGetLastError());
    prof_end(s, MZK_PH_MSM_SEG_COMBINE);
  }
  MZK_TRY(reduce_bucket_sets(buckets, sh.lgB, merged ? 1 : sh.nwin, merged, merged ? 0 : sh.c, wsum, (u32*)d_out, out_partial_xyzz, s));
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// ---- many small commitments against ONE SRS in one pass (grid-batched) ---------------------------------------------------
// The reference's callers commit to hundreds of short polynomials against one `pk` in a loop: one commit_kzg per row
// (das/avail.rs:88-98), per chunk (das/eigenda.rs:92-101), per folded polynomial (algebra/gemini.rs:112-114), and open per
// cell (avail.rs:132).  One such commitment is nothing but latency on this machine (0.135 ms at 2^10 coefficients: two
// launches, 250 CUs idle under a 75-us single-workgroup tail), so the batch is laid out as ONE large bucket problem:
// bucket (j, b) = bucket b of polynomial j, count * 2^(c-1) buckets in all, entries = (window table row, sign) exactly as
// in the single commit.  Then the general pipeline's throughput kernels apply unchanged:
//   k_many_count     workgroup (j, chunk of 1024 coefficients): digits -> LDS histogram -> cnt[j][b][chunk]
//   exclusive scan   over [j][b][chunk]: bucket (j, b) is the concatenation of its chunks' runs
//   k_many_scatter   the same walk again, LDS cursors, entries to their final place
//   (polynomials of <= 1024 coefficients: the three as ONE launch, k_many_sort1, with fixed-capacity entry regions)
//   k_seg_accumulate / k_seg_combine[_heavy]   one lane per fixed-size segment of the sorted entries (mixed additions)
//   k_halve_step*    the wide halving steps, grid.y = polynomial
//   k_reduce_tail_row  one workgroup PER POLYNOMIAL: late steps, weighted sum, affine conversion (wave inversion)
// Every output is the canonical affine point of the same group element the single call returns, hence bit-identical.
constexpr int MANY_CHUNK = 1024;
constexpr int MANY_THREADS = 256;
constexpr int MANY_PER_LANE = MANY_CHUNK / MANY_THREADS;
template <int C>
__global__ __launch_bounds__(MANY_THREADS) void k_many_count(const u32* __restrict__ scalars, size_t n, size_t stride_words, int nch, u32* __restrict__ cnt) {
  constexpr int NB = 1 << (C - 1);
  __shared__ u32 hist[NB];
  const size_t j = blockIdx.x / (unsigned)nch;
  const int ch = (int)(blockIdx.x % (unsigned)nch);
  for (int b = threadIdx.x; b < NB; b += MANY_THREADS) hist[b] = 0;
  __syncthreads();
  const u32* sc = scalars + j * stride_words;
  const size_t lo = (size_t)ch * MANY_CHUNK;
  const size_t hi = (lo + MANY_CHUNK < n) ? lo + MANY_CHUNK : n;
  u32 w[MANY_PER_LANE][8];
#pragma unroll
  for (int k = 0; k < MANY_PER_LANE; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * MANY_THREADS;
    if (i < hi) load_scalar_canonical(sc, i, w[k]);
  }
#pragma unroll
  for (int k = 0; k < MANY_PER_LANE; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * MANY_THREADS;
    if (i < hi) walk_digits_merged<C>(w[k], 0, i, [&](int, u32 key, u32) { counter_inc_agg(hist, key); });
  }
  __syncthreads();
  for (int b = threadIdx.x; b < NB; b += MANY_THREADS) cnt[(j * NB + b) * (size_t)nch + ch] = hist[b];
}
// offs = exclusive scan of cnt (offs[total count] = number of entries).  compact (nch > 1 only): the bucket offsets
// offs[j][b][0] gathered into the dense array k_seg_accumulate reads, compact[nbuckets] = total.
template <int C>
__global__ __launch_bounds__(MANY_THREADS) void k_many_scatter(const u32* __restrict__ scalars, size_t n, size_t stride_words, int nch, size_t table_stride,
                                                                const u32* __restrict__ offs, u32* __restrict__ compact, size_t nbuckets,
                                                                u32* __restrict__ entries) {
  constexpr int NB = 1 << (C - 1);
  __shared__ u32 cursor[NB];
  const size_t j = blockIdx.x / (unsigned)nch;
  const int ch = (int)(blockIdx.x % (unsigned)nch);
  for (int b = threadIdx.x; b < NB; b += MANY_THREADS) {
    const u32 o = offs[(j * NB + b) * (size_t)nch + ch];
    cursor[b] = o;
    if (compact != nullptr && ch == 0) compact[j * NB + b] = o;
  }
  if (compact != nullptr && blockIdx.x == 0 && threadIdx.x == 0) compact[nbuckets] = offs[nbuckets * (size_t)nch];
  __syncthreads();
  const u32* sc = scalars + j * stride_words;
  const size_t lo = (size_t)ch * MANY_CHUNK;
  const size_t hi = (lo + MANY_CHUNK < n) ? lo + MANY_CHUNK : n;
  u32 w[MANY_PER_LANE][8];
#pragma unroll
  for (int k = 0; k < MANY_PER_LANE; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * MANY_THREADS;
    if (i < hi) load_scalar_canonical(sc, i, w[k]);
  }
#pragma unroll
  for (int k = 0; k < MANY_PER_LANE; k++) {
    const size_t i = lo + threadIdx.x + (size_t)k * MANY_THREADS;
    if (i < hi) walk_digits_merged<C>(w[k], table_stride, i, [&](int, u32 key, u32 payload) { entries[counter_inc_agg(cursor, key)] = payload; });
  }
}

// Polynomials of at most MANY_CHUNK coefficients: count, scan and scatter in ONE launch.  Workgroup j owns polynomial j and the entry
// region [j CAP, (j + 1) CAP), CAP = n NWIN: LDS histogram (first walk), exclusive scan inside the workgroup, bucket offsets
// j CAP + prefix (no global scan: the regions have a fixed size), placement through LDS cursors (second walk), and the unused tail
// of the region -- zero digits emit nothing -- filled with sentinels, which k_seg_accumulate<.., SENT> skips and bucket_end (tails[j]) keeps
// out of the last bucket's sum.  One coefficient per lane (1024 lanes: at 256 polynomials the kernel is one workgroup per CU, i.e.
// latency) and the sorted region staged in LDS (<= 128 KiB) and copied out as one stream: **19 us** at 256 x 2^10 against 17 + 10 +
// 35 us and three more launch gaps for the count / scan / scatter form (placed directly into global memory it was 42 us: 32
// isolated four-byte stores per lane).  Longer polynomials keep the three-launch form: their buckets span several chunks.
constexpr int SORT1_THREADS = 1024;
constexpr int SORT1_PER_LANE = MANY_CHUNK / SORT1_THREADS;
template <int C>
__global__ __launch_bounds__(SORT1_THREADS) void k_many_sort1(const u32* __restrict__ scalars, size_t n, size_t stride_words, size_t table_stride, u32 cap,
                                                              u32* __restrict__ offsets, size_t npoly, u32* __restrict__ entries, u32* __restrict__ tails) {
  constexpr int NB = 1 << (C - 1);
  constexpr int PER = (NB + SORT1_THREADS - 1) / SORT1_THREADS;      // counters per lane in the scan
  __shared__ u32 hist[NB];
  __shared__ u32 scan[SORT1_THREADS];
  extern __shared__ u32 stage[];                                   // [cap] the polynomial's entries, bucket-sorted
  const size_t j = blockIdx.x;
  const int tid = threadIdx.x;
  for (int b = tid; b < NB; b += SORT1_THREADS) hist[b] = 0;
  __syncthreads();
  const u32* sc = scalars + j * stride_words;
  u32 w[SORT1_PER_LANE][8];
#pragma unroll
  for (int k = 0; k < SORT1_PER_LANE; k++) {
    const size_t i = tid + (size_t)k * SORT1_THREADS;
    if (i < n) load_scalar_canonical(sc, i, w[k]);
  }
#pragma unroll
  for (int k = 0; k < SORT1_PER_LANE; k++) {
    const size_t i = tid + (size_t)k * SORT1_THREADS;
    if (i < n) walk_digits_merged<C>(w[k], 0, i, [&](int, u32 key, u32) { counter_inc_agg(hist, key); });
  }
  __syncthreads();
  u32 local = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) { const int b = tid * PER + q; if (b < NB) local += hist[b]; }
  u32 total;
  u32 run = block_excl_scan<SORT1_THREADS>(local, scan, &total);
  const u32 base = (u32)(j * cap);
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int b = tid * PER + q;
    if (b < NB) { const u32 c = hist[b]; hist[b] = run; offsets[j * NB + b] = base + run; run += c; }       // hist becomes the cursor
  }
  if (j + 1 == npoly && tid == 0) offsets[npoly * NB] = (u32)(npoly * cap);
  if (tid == 0) tails[j] = base + total;          // where the sentinels begin: bucket_end
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SORT1_PER_LANE; k++) {
    const size_t i = tid + (size_t)k * SORT1_THREADS;
    if (i < n) walk_digits_merged<C>(w[k], table_stride, i, [&](int, u32 key, u32 payload) { stage[counter_inc_agg(hist, key)] = payload; });
  }
  __syncthreads();
  // the sorted region leaves LDS as one contiguous stream (placed directly, every lane's 32 four-byte stores went to 32 different
  // lines: 42 us for the kernel; staged: the copy-out is 128 KiB of consecutive words), sentinels behind it
  for (u32 e = tid; e < cap; e += SORT1_THREADS) entries[base + e] = (e < total) ? stage[e] : MANY_SENTINEL;
}

bool msm_many_supported(int window_bits) { return window_bits == 8 || (window_bits >= 10 && window_bits <= 13); }

// `count` MSMs of n scalars each (polynomial j at d_scalars + j * stride_elems * 32 bytes) against the window tables of one SRS
// handle (c-bit windows, msm_many_supported(c)); d_out: count affine points (ABI form), 64 bytes apart.
int msm_many_dev_impl(const void* d_scalars, size_t n, size_t stride_elems, size_t count, const void* d_tables, int c, size_t table_stride, void* d_out,
                      hipStream_t s) {
  if (count == 0) return MZK_OK;
  if (!d_out || ((!d_scalars || !d_tables) && n)) { set_error("msm_many: null pointer"); return MZK_E_ARG; }
  if (!msm_many_supported(c)) { set_error("msm_many: window width %d has no grid-batched path", c); return MZK_E_ARG; }
  if (n == 0) {  // empty polynomials -> points at infinity (polynomial.rs:160)
    MZK_HIP(hipMemsetAsync(d_out, 0, count * 64, s));
    return MZK_OK;
  }
  const int nwin = msm_table_windows(c), lgB = c - 1;
  const size_t NB = (size_t)1 << lgB;
  if ((size_t)nwin * table_stride > ((size_t)1 << 31)) { set_error("msm_many: table rows exceed the 31-bit point references"); return MZK_E_ARG; }
  const int nch = (int)((n + MANY_CHUNK - 1) / MANY_CHUNK);
  // one pass handles at most 2^21 buckets and 2^22 coefficients (entries, partial slots and buckets stay below ~1.5 GiB of workspace)
  size_t per_pass = ((size_t)1 << 21) / NB;
  const size_t by_coefs = (((size_t)1 << 22) + n - 1) / n;
  if (by_coefs < per_pass) per_pass = by_coefs;
  if (per_pass < 1) per_pass = 1;
  for (size_t first = 0; first < count; first += per_pass) {
    const size_t cnt = (count - first < per_pass) ? count - first : per_pass;
    const u32* sc = (const u32*)d_scalars + first * stride_elems * 8;
    u32* out = (u32*)d_out + first * 16;
    const size_t NBtot = cnt * NB, ncnt = NBtot * (size_t)nch;
    const size_t E_max = cnt * n * (size_t)nwin;
    const size_t resident_lanes = (size_t)ctx().num_cu * 4 * 4 * 64;
    size_t seg_sz = (E_max + resident_lanes - 1) / resident_lanes;
    if (seg_sz < 16) seg_sz = 16;
    const u32 seg = (u32)seg_sz;
    const size_t T = (E_max + seg_sz - 1) / seg_sz;
    const size_t nslots = T + NBtot + 1;
    const size_t max_heavy = (T + NBtot) / HEAVY_SLOTS_SORT1 + 1;
    const size_t heavy_words = heavy_total_words(max_heavy);
    u32 *offs, *compact, *entries, *scan_tmp, *buckets, *slots;
    MZK_TRY(ws_get(WS_MSM_COUNTS, (ncnt + 1) * 4, (void**)&offs));
    MZK_TRY(ws_get(WS_MSM_ENTRIES, E_max * 4, (void**)&entries));
    MZK_TRY(ws_get(WS_MSM_SCAN, scan_scratch_words(ncnt) * 4, (void**)&scan_tmp));
    MZK_TRY(ws_get(WS_MSM_BUCKETS, NBtot * 128, (void**)&buckets));
    MZK_TRY(ws_get(WS_MSM_SLOTS, nslots * SLOT_WORDS * 4 + heavy_words * 4, (void**)&slots));
    compact = offs;
    u32* tails = nullptr;                   // one-kernel sort only: end of every polynomial's real entries (bucket_end)
    if (nch > 1) MZK_TRY(ws_get(WS_MSM_OFFSETS, (NBtot + 1) * 4, (void**)&compact));
    else MZK_TRY(ws_get(WS_MSM_OFFSETS, cnt * 4, (void**)&tails));
    u32* heavy = slots + nslots * SLOT_WORDS;
    prof_begin(s, MZK_PH_MSM_SORT);
    MZK_HIP(hipMemsetAsync(heavy, 0, HEAVY_CLEAR_BYTES, s));
    const unsigned nwg = (unsigned)(cnt * (size_t)nch);
    const bool one_kernel_sort = nch == 1;      // polynomials of <= 1024 coefficients: fixed-capacity regions, k_many_sort1
    if (one_kernel_sort) {                      // its staged region is up to 128 KiB of LDS: the attribute, once per context
      bool& done = ctx().attr_done[ATTR_MANY_SORT1];
      if (!done) {
        MZK_HIP(hipFuncSetAttribute((const void*)k_many_sort1<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        MZK_HIP(hipFuncSetAttribute((const void*)k_many_sort1<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        MZK_HIP(hipFuncSetAttribute((const void*)k_many_sort1<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        MZK_HIP(hipFuncSetAttribute((const void*)k_many_sort1<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        MZK_HIP(hipFuncSetAttribute((const void*)k_many_sort1<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        done = true;
      }
    }
#define MZK_MANY_CASE(C) case C:                                                                                                                    \
      if (one_kernel_sort) {                                                                                                                         \
        hipLaunchKernelGGL((k_many_sort1<C>), dim3(nwg), dim3(SORT1_THREADS), n * (size_t)nwin * 4, s, sc, n, stride_elems * 8, table_stride,       \
                           (u32)(n * (size_t)nwin), offs, cnt, entries, tails);                                                                           \
        break;                                                                                                                                       \
      }                                                                                                                                              \
      hipLaunchKernelGGL((k_many_count<C>), dim3(nwg), dim3(MANY_THREADS), 0, s, sc, n, stride_elems * 8, nch, offs);                                \
      MZK_TRY(launch_exclusive_scan((const u32*)offs, offs, ncnt, scan_tmp, s));                                                                     \
      hipLaunchKernelGGL((k_many_scatter<C>), dim3(nwg), dim3(MANY_THREADS), 0, s, sc, n, stride_elems * 8, nch, table_stride, (const u32*)offs,     \
                         nch > 1 ? compact : (u32*)nullptr, NBtot, entries);                                                                         \
      break;
    switch (c) { MZK_MANY_CASE(8) MZK_MANY_CASE(10) MZK_MANY_CASE(11) MZK_MANY_CASE(12) MZK_MANY_CASE(13) }
#undef MZK_MANY_CASE
    MZK_HIP(hipGetLastError());
    prof_end(s, MZK_PH_MSM_SORT);
    prof_begin(s, MZK_PH_MSM_ACCUMULATE);
    // (the one-kernel sort's regions have a fixed capacity, sentinels included: the host's segment length stands there)
    const u32 t_max = one_kernel_sort ? 0u : (u32)T;
    if (one_kernel_sort)
      hipLaunchKernelGGL((k_seg_accumulate<false, true>), dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, (const u32*)d_tables, (const u32*)compact, (const u32*)entries,
                         slots, NBtot, seg, t_max);
    else
      hipLaunchKernelGGL(k_seg_accumulate<false>, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, (const u32*)d_tables, (const u32*)compact, (const u32*)entries,
                         slots, NBtot, seg, t_max);
    prof_end(s, MZK_PH_MSM_ACCUMULATE);
    prof_begin(s, MZK_PH_MSM_SEG_COMBINE);
    // ONE quad per bucket.  Two or four quads per bucket, each taking every 2nd / 4th partial and folding through shuffles (the "tree" of
    // VERDICT r05 #5 and of round 5's DESIGN 9a iii), were built and measured: slower at every shape -- 64 x 2^12 segment combine 0.111 ->
    // 0.149 / 0.167 ms, 16 x 2^14 0.112 -> 0.130 / 0.168, 256 x 2^10 0.073 -> 0.092 / 0.135 (profiles/round6_seg_combine_quads_per_bucket_ab.txt):
    // the kernel is bound by the instructions of its additions, not by the length of a bucket's chain, and the fold adds some.
    const unsigned cgrid = (unsigned)((4 * NBtot + 127) / 128);
#ifdef MZK_TUNING
    static const int env_qpb = tune_int("MZK_COMBINE_QPB", 1);          // tuning build: 2 / 4 = the measured-and-dropped forms
    if (env_qpb == 4)
      hipLaunchKernelGGL(k_seg_combine<4>, dim3(4 * cgrid), dim3(128), 0, s, slots, (const u32*)compact, buckets, NBtot, seg, heavy, (const u32*)tails, lgB, t_max);
    else if (env_qpb == 2)
      hipLaunchKernelGGL(k_seg_combine<2>, dim3(2 * cgrid), dim3(128), 0, s, slots, (const u32*)compact, buckets, NBtot, seg, heavy, (const u32*)tails, lgB, t_max);
    else
#endif
      hipLaunchKernelGGL(k_seg_combine<1>, dim3(cgrid), dim3(128), 0, s, slots, (const u32*)compact, buckets, NBtot, seg, heavy, (const u32*)tails, lgB, t_max);
    hipLaunchKernelGGL(k_seg_combine_heavy, dim3(HEAVY_GRID), dim3(HEAVY_THREADS), 0, s, (const u32*)slots, (const u32*)compact, buckets, heavy, max_heavy);
    MZK_HIP(hipGetLastError());
    prof_end(s, MZK_PH_MSM_SEG_COMBINE);
    prof_begin(s, MZK_PH_MSM_REDUCE);
    int t_start = 0;
    MZK_TRY(launch_halving_steps(buckets, lgB, (int)cnt, row_tail_max_ops(), &t_start, s));
    MZK_TRY(launch_reduce_tail_row(buckets, lgB, t_start, (int)cnt, out, 1, s));
    MZK_HIP(hipGetLastError());
    prof_end(s, MZK_PH_MSM_REDUCE);
  }
  return MZK_OK;
}

// ---- the same batch over DIRECT tables: no buckets at all ------------------------------------------------------------------
// D[(w n + i) M + m] = (m + 1) 2^(c w) P_i for every magnitude a signed c-bit digit can take (M = 2^(c-1)): a commitment is then
// sum over (i, w) of +-D[w][i][|d| - 1] -- n (254 / c + 1) mixed additions in ANY order, so there is no digit sort, no bucket
// accumulation by segments, no segment combine, no bucket reduction.  Affordable only because the polynomials are short and the
// HBM is large: 0.85 GiB for a 1024-power SRS at 10-bit windows (26 x 1024 x 512 points), built once per handle on request
// (mzk_srs_build_direct).  Two launches for the whole batch:
//   k_direct_accumulate  workgroup = 256 coefficients of one polynomial, one coefficient per lane: its digits (the long addition
//                        of walk_digits_merged, then one window per trip), one gathered row and one mixed addition per window,
//                        then the 256 partials -> one by the tree of the small-commit path (quads, the last levels row additions)
//   k_fold_partials_row  one wave per polynomial: its ceil(n / 256) partials, then the affine conversion (wave inversion)
constexpr int DIRECT_THREADS = 256;
size_t msm_direct_bytes(size_t n, int c) { return (size_t)msm_table_windows(c) * n * ((size_t)1 << (c - 1)) * 64; }
// multiples 1 .. M of `rows` table rows (affine Montgomery) as XYZZ records, row-major: out[(r M + m)] = (m + 1) T[r]
__global__ __launch_bounds__(64) void k_direct_chain(const u32* __restrict__ rows_mont, size_t rows, int M, u32* __restrict__ out_xyzz) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  u32 w[16];
  load_words8(rows_mont + r * 16, w);
  load_words8(rows_mont + r * 16 + 8, w + 8);
  const bool inf = affine_words_is_inf(w);
  const Affine a = affine_load_mont(w);
  Xyzz acc = xyzz_inf();
  for (int m = 0; m < M; m++) {
    if (!inf) acc = xyzz_madd_signed_with<FeCpp>(acc, a, false);
    xyzz_gstore(out_xyzz, r * (size_t)M + m, acc);
  }
}
int msm_build_direct(const void* d_points_mont, size_t n, int c, void* d_direct, hipStream_t s) {
  if (n == 0) return MZK_OK;
  const int nwin = msm_table_windows(c), M = 1 << (c - 1);
  const size_t rows = (size_t)nwin * n;
  // window tables T[w][i] = 2^(c w) P_i of this width (scratch), then the multiples of every row in slices of <= 2^21 records
  u32 *tables, *tmp;
  MZK_TRY(ws_get(WS_MISC_C, rows * 64, (void**)&tables));
  MZK_TRY(msm_build_tables(d_points_mont, n, tables, c, s));
  size_t slice = ((size_t)1 << 21) / (size_t)M;
  if (slice < 64) slice = 64;
  if (slice > rows) slice = rows;
  MZK_TRY(ws_get(WS_MISC_D, slice * (size_t)M * 128, (void**)&tmp));
  for (size_t r0 = 0; r0 < rows; r0 += slice) {
    const size_t cnt = (rows - r0 < slice) ? rows - r0 : slice;
    hipLaunchKernelGGL(k_direct_chain, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, s, (const u32*)tables + r0 * 16, cnt, M, tmp);
    MZK_HIP(hipGetLastError());
    MZK_TRY(xyzz_batch_to_affine(tmp, cnt * (size_t)M, (u32*)d_direct + r0 * (size_t)M * 16, true, s));
  }
  return MZK_OK;
}
// Work item t of a polynomial = (coefficient t / NWIN, window t % NWIN).  A polynomial's n NWIN items are cut into `parts`
// contiguous ranges, one workgroup each, lane l taking items first + l, first + l + 256, ...: the host picks `parts` so that the
// grid is about ONE round of resident workgroups whatever count and n are (a lane per coefficient would be 26 additions long
// and, at 256 x 2^10, 1.33 rounds: measured 0.60 ms where the additions themselves are 0.48).  Every item recomputes the
// biased scalar of its coefficient (one 256-bit addition: ~1 % of the mixed addition it feeds; consecutive lanes mostly share
// the coefficient, so the loads are broadcasts), requests the NEXT item's row before adding the current one, and the 256
// partials end in the tree of the small-commit path.
template <int C>
__device__ __forceinline__ bool direct_item(const u32* __restrict__ sc, size_t t, size_t table_stride, const u32* __restrict__ direct, const u32** rec, bool* neg) {
  constexpr int NWIN = 254 / C + 1;
  constexpr u32 HALF = 1u << (C - 1), MASKC = (1u << C) - 1u;
  const size_t i = t / NWIN;
  const int win = (int)(t - i * NWIN);
  u32 w[8], tw[10];
  load_scalar_canonical(sc, i, w);
  u64 cy = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { cy += (u64)w[k] + digit_bias_word(C, k); tw[k] = (u32)cy; cy >>= 32; }
  tw[8] = (u32)cy + digit_bias_word(C, 8);
  tw[9] = 0;
  const int bit = win * C, k = bit >> 5, sft = bit & 31;
  u32 lo = tw[0], hi = tw[1];
#pragma unroll
  for (int q = 1; q < 9; q++) { lo = (k == q) ? tw[q] : lo; hi = (k == q) ? tw[q + 1] : hi; }
  const u32 v = (u32)((((u64)hi << 32) | lo) >> sft) & MASKC;       // digit + HALF - 1
  if (v == HALF - 1u) return false;                                  // digit 0
  *neg = v < HALF - 1u;
  const u32 mag = *neg ? (HALF - 1u) - v : v - (HALF - 1u);
  *rec = direct + (((size_t)win * table_stride + i) * HALF + (mag - 1u)) * 16;
  return true;
}
template <int C>
__global__ __launch_bounds__(DIRECT_THREADS) void k_direct_accumulate(const u32* __restrict__ scalars, size_t n, size_t stride_words, int parts, size_t table_stride,
                                                                       const u32* __restrict__ direct, u32* __restrict__ slots) {
  constexpr int NWIN = 254 / C + 1;
  const size_t j = blockIdx.x / (unsigned)parts;
  const int part = (int)(blockIdx.x % (unsigned)parts);
  const u32* sc = scalars + j * stride_words;
  const size_t items = n * NWIN;
  const size_t per = (items + parts - 1) / parts;
  const size_t first = (size_t)part * per;
  const size_t last = (first + per < items) ? first + per : items;
  Xyzz acc = xyzz_inf();
  u32 pw[16];
  const u32* rec = nullptr;
  bool neg = false;
  size_t t = first + threadIdx.x;
  bool have = (t < last) && direct_item<C>(sc, t, table_stride, direct, &rec, &neg);
  if (have) { load_words8(rec, pw); load_words8(rec + 8, pw + 8); }
  while (t < last) {
    const bool cur_have = have, cur_neg = neg;
    u32 cw[16];
#pragma unroll
    for (int q = 0; q < 16; q++) cw[q] = pw[q];
    t += DIRECT_THREADS;
    have = (t < last) && direct_item<C>(sc, t, table_stride, direct, &rec, &neg);
    if (have) { load_words8(rec, pw); load_words8(rec + 8, pw + 8); }
    if (cur_have && !affine_words_is_inf(cw)) acc = xyzz_madd_signed_with<FeAsm>(acc, affine_load_mont(cw), cur_neg);
  }
  xyzz_gstore_raw(slots, (size_t)blockIdx.x * DIRECT_THREADS + threadIdx.x, acc);
}
// one workgroup per polynomial: lane l sums the partials of lane l of its `parts` accumulate workgroups, the 256 sums go through
// the tree of the small-commit path, and wave 0 converts the result to the affine point
__global__ __launch_bounds__(DIRECT_THREADS) void k_direct_finish(const u32* __restrict__ slots, int parts, u32* __restrict__ scratch, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 sh[DIRECT_THREADS * 32];
  const size_t j = blockIdx.x;
  Xyzz acc = xyzz_gload_raw(slots, (j * parts) * DIRECT_THREADS + threadIdx.x);
  for (int p = 1; p < parts; p++) acc = xyzz_add_with<FeAsm>(acc, xyzz_gload_raw(slots, (j * parts + p) * DIRECT_THREADS + threadIdx.x));
  small_tree_store<DIRECT_THREADS>(sh, acc, DIRECT_THREADS, j, scratch);          // leaves the sum in sh[0 .. 32) as well
  __syncthreads();
  if (threadIdx.x < 64) wave_store_affine(sh, out + j * 16);
}
// more than DIRECT_FINISH_PARTS partial sets per polynomial (few, long polynomials): workgroup (j, g) of `groups` per polynomial
// sums the sets g, g + groups, ... lane by lane first, so that no lane ever chains more than ~sqrt(parts) additions
constexpr int DIRECT_FINISH_PARTS = 8;
__global__ __launch_bounds__(DIRECT_THREADS) void k_direct_fold(const u32* __restrict__ slots, int parts, int groups, u32* __restrict__ folded) {
  const size_t j = blockIdx.x / (unsigned)groups;
  const int g = (int)(blockIdx.x % (unsigned)groups);
  Xyzz acc = xyzz_gload_raw(slots, (j * parts + g) * DIRECT_THREADS + threadIdx.x);
  for (int p = g + groups; p < parts; p += groups) acc = xyzz_add_with<FeAsm>(acc, xyzz_gload_raw(slots, (j * parts + p) * DIRECT_THREADS + threadIdx.x));
  xyzz_gstore_raw(folded, (size_t)blockIdx.x * DIRECT_THREADS + threadIdx.x, acc);
}
int msm_direct_many_dev_impl(const void* d_scalars, size_t n, size_t stride_elems, size_t count, const void* d_direct, int c, size_t table_stride, void* d_out,
                             hipStream_t s) {
  if (count == 0) return MZK_OK;
  if (!d_out || ((!d_scalars || !d_direct) && n)) { set_error("msm_direct_many: null pointer"); return MZK_E_ARG; }
  if (n == 0) {
    MZK_HIP(hipMemsetAsync(d_out, 0, count * 64, s));
    return MZK_OK;
  }
  const size_t items = n * (size_t)msm_table_windows(c);
  const size_t per_pass = (size_t)1 << 12;                           // polynomials per pass: cnt x parts <= ~4096 workgroups of lane partials (36 KiB each: <= 150 MiB)
  for (size_t first = 0; first < count; first += per_pass) {
    const size_t cnt = (count - first < per_pass) ? count - first : per_pass;
    // workgroups per polynomial: one round of resident workgroups over the batch (three 256-lane workgroups per CU at the
    // kernel's VGPR count), at least one, at most one per 256 items, and at most 64 (k_direct_fold / k_direct_finish add them)
    static const int parts_mul = tune_int("MZK_DIRECT_PARTS_MUL", 1);       // tuning build: rounds of workgroups per launch
    const size_t resident = (size_t)ctx().num_cu * 3 * (size_t)(parts_mul > 0 ? parts_mul : 1);
    size_t parts = (resident + cnt / 2) / cnt;
    const size_t max_parts = (items + DIRECT_THREADS - 1) / DIRECT_THREADS;
    if (parts > max_parts) parts = max_parts;
    if (parts > 64) parts = 64;
    if (parts < 1) parts = 1;
    size_t groups = 0;
    if (parts > (size_t)DIRECT_FINISH_PARTS) for (groups = 2; groups * groups < parts; groups++) {}
    u32 *slots, *scratch;
    MZK_TRY(ws_get(WS_MSM_SLOTS, cnt * (parts + groups) * DIRECT_THREADS * SLOT_WORDS * 4, (void**)&slots));
    MZK_TRY(ws_get(WS_MSM_BUCKETS, cnt * 128, (void**)&scratch));
    const u32* sc = (const u32*)d_scalars + first * stride_elems * 8;
    const unsigned nwg = (unsigned)(cnt * parts);
    prof_begin(s, MZK_PH_MSM_ACCUMULATE);
#define MZK_DIRECT_CASE(C) case C: hipLaunchKernelGGL((k_direct_accumulate<C>), dim3(nwg), dim3(DIRECT_THREADS), 0, s, sc, n, stride_elems * 8, (int)parts, table_stride, (const u32*)d_direct, slots); break;
    switch (c) { MZK_DIRECT_CASE(8) MZK_DIRECT_CASE(9) MZK_DIRECT_CASE(10) MZK_DIRECT_CASE(11) MZK_DIRECT_CASE(12)
      default: set_error("msm_direct_many: no kernel for %d-bit direct tables", c); return MZK_E_ARG; }
#undef MZK_DIRECT_CASE
    MZK_HIP(hipGetLastError());
    prof_end(s, MZK_PH_MSM_ACCUMULATE);
    prof_begin(s, MZK_PH_MSM_REDUCE);
    if (groups) {
      u32* folded = slots + cnt * parts * DIRECT_THREADS * SLOT_WORDS;
      hipLaunchKernelGGL(k_direct_fold, dim3((unsigned)(cnt * groups)), dim3(DIRECT_THREADS), 0, s, (const u32*)slots, (int)parts, (int)groups, folded);
      hipLaunchKernelGGL(k_direct_finish, dim3((unsigned)cnt), dim3(DIRECT_THREADS), 0, s, (const u32*)folded, (int)groups, scratch, (u32*)d_out + first * 16);
    } else {
      hipLaunchKernelGGL(k_direct_finish, dim3((unsigned)cnt), dim3(DIRECT_THREADS), 0, s, (const u32*)slots, (int)parts, scratch, (u32*)d_out + first * 16);
    }
    MZK_HIP(hipGetLastError());
    prof_end(s, MZK_PH_MSM_REDUCE);
  }
  return MZK_OK;
}
bool srs_many_capable(const mzk_srs* srs) { return srs->d_direct != nullptr || (srs->has_tables && srs->sets == 1 && msm_many_supported(srs->window_bits)); }
// Window width of the bucket pass by polynomial length (same-box sweep, profiles/round5_many_commit_widths.txt: 16 x 2^14 0.884 ms at
// 10 bits, 0.707 at 12; 32 x 2^13 0.842 / 0.713; 64 x 2^12 0.730 / 0.752; 128 x 2^11 0.755 / 1.236): 12 bits from 2^13 coefficients on.
constexpr size_t MANY_WIDE_FROM = (size_t)1 << 13;
constexpr int MANY_WIDE_BITS = 12;
int msm_many_srs(const mzk_srs* srs, const void* d_scalars, size_t n, size_t stride_elems, size_t count, void* d_out, hipStream_t s) {
  if (srs->d_direct) return msm_direct_many_dev_impl(d_scalars, n, stride_elems, count, srs->d_direct, srs->direct_bits, srs->n, d_out, s);
  if (n >= MANY_WIDE_FROM && srs->has_tables && srs->sets == 1 && srs->window_bits < MANY_WIDE_BITS) {
    if (!srs->d_tables_wide && srs->wide_bits == 0) {       // once per handle: row 0 of its own tables are the prepared points
      const size_t bytes = (size_t)msm_table_windows(MANY_WIDE_BITS) * srs->n * 64;
      const size_t own = (size_t)srs->n * 64 * srs->table_rows() + srs->direct_bytes;
      void* t = nullptr;
      // the caller's table budget (mzk_set_table_budget) covers these tables too; a refusal -- budget or device -- is remembered on the
      // handle (wide_bits = -1), so that later batches do not repeat a failing hipMalloc and the idle-workspace release it triggers,
      // and the message of the refused allocation does not stay behind on a call that returns MZK_OK
      const bool within_budget = table_budget_bytes() == 0 || own + bytes <= table_budget_bytes();
      if (within_budget && dev_alloc(&t, bytes, "wide window tables of the grid-batched pass") == MZK_OK) {
        const int rc = msm_build_tables(srs->d_points_mont, srs->n, t, MANY_WIDE_BITS, s);
        if (rc != MZK_OK) { (void)hipFree(t); return rc; }
        srs->d_tables_wide = t; srs->wide_bits = MANY_WIDE_BITS; srs->wide_bytes = bytes;
      } else {
        srs->wide_bits = -1;         // no memory (or no budget) for them: the handle's own tables serve (slower, same points)
        if (within_budget) clear_error();
      }
    }
    if (srs->d_tables_wide) return msm_many_dev_impl(d_scalars, n, stride_elems, count, srs->d_tables_wide, srs->wide_bits, srs->n, d_out, s);
  }
  return msm_many_dev_impl(d_scalars, n, stride_elems, count, srs->d_points_mont, srs->window_bits, srs->n, d_out, s);
}

int msm_fold_partials_impl(const void* d_partials, int count, void* d_out_xy, hipStream_t s) {
  if (!d_partials || !d_out_xy || count < 0) { set_error("fold_partials: bad argument"); return MZK_E_ARG; }
#ifdef MZK_TUNING
  if (!row_tails()) return launch_fold_partials((const u32*)d_partials, count, (u32*)d_out_xy, s);
#endif
  MZK_TRY(launch_fold_partials_row((const u32*)d_partials, count, (u32*)d_out_xy, s));
  return MZK_OK;
}

}  // namespace mzk

// ---- deterministic synthetic inputs (bit-identical to oracle/mzk_oracle.c orc_synth_*) -------------------
namespace mzk {
__device__ __forceinline__ u64 splitmix64(u64& s) {
  u64 z = (s += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
// canonical element `index` of stream `seed` as NW words
template <class P> __device__ void synth_words(u64 seed, u64 index, u32* w) {
  constexpr int NL = P::NW / 2;
  // modulus as 32-bit words, from the 29-bit limbs
  u32 pw[P::NW];
  {
    Fe<P> pm;
    for (int i = 0; i < P::L; i++) pm.l[i] = P::P[i];
    fe_pack<P>(pm, pw);
  }
  for (u64 attempt = 0;; attempt++) {
    u64 s = seed ^ (index * 0xd1342543de82ef95ULL) ^ (attempt * 0xa0761d6478bd642fULL);
    for (int i = 0; i < NL; i++) {
      u64 v = splitmix64(s);
      w[2 * i] = (u32)v;
      w[2 * i + 1] = (u32)(v >> 32);
    }
    constexpr int topbits = P::BITS - 64 * (NL - 1);
    if (topbits < 64) {
      const u64 mask = (((u64)1 << topbits) - 1);
      w[P::NW - 2] &= (u32)mask;
      w[P::NW - 1] &= (u32)(mask >> 32);
    }
    bool lt = false, decided = false;
    for (int i = P::NW - 1; i >= 0 && !decided; i--) {
      if (w[i] != pw[i]) { lt = w[i] < pw[i]; decided = true; }
    }
    if (decided && lt) return;
  }
}
template <class P> __global__ void k_synth_field(u64 seed, size_t n, u32* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[P::NW];
  synth_words<P>(seed, i, w);
  for (int k = 0; k < P::NW; k++) out[i * P::NW + k] = w[k];
}
// try-and-increment G1 points: x = synth(Fq, seed, i) (+1 until x^3+3 is a non-zero square),
// y = (x^3+3)^((q+1)/4)
__global__ void k_synth_g1(u64 seed, size_t n, u32* __restrict__ out) {
  typedef FqParams P;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  synth_words<P>(seed, i, w);
  // exponent (q+1)/4 as words: q = 3 mod 4, so (q+1)/4 = (q >> 2) + 1
  u32 e[8];
  {
    Fe<P> pm;
    for (int k = 0; k < P::L; k++) pm.l[k] = P::P[k];
    u32 qw[8];
    fe_pack<P>(pm, qw);
    for (int k = 0; k < 8; k++) e[k] = (qw[k] >> 2) | ((k + 1 < 8) ? (qw[k + 1] << 30) : 0u);
    u32 c = 1;
    for (int k = 0; k < 8 && c; k++) { e[k] += c; c = (e[k] == 0) ? 1u : 0u; }
  }
  Fe<P> b3;
  for (int k = 0; k < P::L; k++) b3.l[k] = FQ_B_MONT[k];
  Fe<P> x = fe_reduce<P>(fe_to_mont<P>(fe_unpack<P>(w)));
  Fe<P> one = fe_one<P>();
  for (;;) {
    Fe<P> rhs = fe_reduce<P>(fe_add<P>(fe_mul<P>(fe_sqr<P>(x), x), b3));
    Fe<P> y = fe_pow_words<P>(rhs, e, 8);
    Fe<P> y2 = fe_reduce<P>(fe_sqr<P>(y));
    if (fe_eq_canon<P>(y2, rhs) && !fe_is_zero_canon<P>(rhs)) {
      Affine a;
      a.x = x;
      a.y = fe_reduce<P>(y);
      u32 o[16];
      affine_store_plain(a, o);
      for (int k = 0; k < 16; k++) out[i * 16 + k] = o[k];
      return;
    }
    x = fe_reduce<P>(fe_add<P>(x, one));
  }
}
int synth_field_impl(int fid, u64 seed, size_t n, void* d_out, hipStream_t s) {
  if (n == 0) return MZK_OK;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (fid == MZK_FIELD_FR) hipLaunchKernelGGL((k_synth_field<FrParams>), dim3(blocks), dim3(256), 0, s, seed, n, (u32*)d_out);
  else if (fid == MZK_FIELD_FQ) hipLaunchKernelGGL((k_synth_field<FqParams>), dim3(blocks), dim3(256), 0, s, seed, n, (u32*)d_out);
  else if (fid == MZK_FIELD_M128) hipLaunchKernelGGL((k_synth_field<M128Params>), dim3(blocks), dim3(256), 0, s, seed, n, (u32*)d_out);
  else { set_error("synth: bad field id %d", fid); return MZK_E_ARG; }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int synth_g1_impl(u64 seed, size_t n, void* d_out, hipStream_t s) {
  if (n == 0) return MZK_OK;
  hipLaunchKernelGGL(k_synth_g1, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, seed, n, (u32*)d_out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
}  // namespace mzk
