// mzk_kzg.hip -- the non-MSM parts of KZG setup / open on gfx950.
//
// setup_kzg (myzkp/src/modules/algebra/kzg.rs:27-40, G1 part): powers[i] = alpha^i * g1.  The reference
// runs max_d + 1 independent double-and-add scalar multiplications; here a fixed-base table
// T[w][d] = d * 2^(8 w) * g1 (w < 32, d < 256) turns each into 32 mixed additions.
//
// open_kzg (kzg.rs:61-72): y = f(u) (Polynomial::eval, polynomial.rs:120-128) and the quotient
// (f - y) / (X - u) (div_rem_ref, polynomial.rs:371-405).  Dividing by a monic linear factor is
// synthetic division: b_{n-1} = c_{n-1}, b_i = c_i + u b_{i+1}; then y = b_0 and q_j = b_{j+1}.  The
// suffix recurrence is solved by recursive chunking (chunk value at u, then the same recurrence on
// the chunk values with base u^K), so it is parallel at every level.  The witness MSM then runs in
// mzk_msm.hip on q.
#include "mzk_common.h"
#include "mzk_ec.h"

namespace mzk {
// fixed-base table cache of kzg_setup_g1_dev, one entry per context (the tables live in that context's workspace, the
// event on its device); kzg_release_cache() drops the current context's entry -- mzk_shutdown calls it, so a re-init that
// puts another device at the same index starts clean.
static struct { uint64_t g[8]; uint64_t gen; bool have8, have16; hipEvent_t ready; } g_fb_cache[MZK_MAX_CTX] = {};
void kzg_release_cache() {
  auto& c = g_fb_cache[ctx().index];
  if (c.ready) (void)hipEventDestroy(c.ready);
  memset(&c, 0, sizeof c);
}

struct Words8k { u32 w[8]; };
struct Words16k { u32 w[16]; };

__device__ __forceinline__ void ld8(const u32* __restrict__ g, u32* w) {
  const uint4* p4 = reinterpret_cast<const uint4*>(g);
  uint4 a = p4[0], b = p4[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void st8(u32* __restrict__ g, const u32* w) {
  uint4* p4 = reinterpret_cast<uint4*>(g);
  p4[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p4[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// ---- fixed-base table ---------------------------------------------------------------------------------
// bases[w] = 2^(8 w) * g1, affine Montgomery (16 words); infinity base -> zeros
__global__ void k_fb_bases(Words16k g1_plain, u32* __restrict__ bases) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= 32) return;
  u32 o[16];
  for (int i = 0; i < 16; i++) o[i] = 0;
  if (!affine_words_is_inf(g1_plain.w)) {
    Xyzz p = xyzz_from_affine(affine_load_plain(g1_plain.w));
    for (int d = 0; d < 8 * w; d++) p = xyzz_dbl(p);
    Affine a;
    if (xyzz_to_affine(p, &a)) affine_store_mont(a, o);
  }
  st8(bases + 16 * w, o);
  st8(bases + 16 * w + 8, o + 8);
}
// table[w][d] = d * bases[w], d = 0..255 (d = 0 and infinity -> zeros), affine Montgomery
__global__ void k_fb_table(const u32* __restrict__ bases, u32* __restrict__ table) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 32 * 256) return;
  const int w = t >> 8, d = t & 255;
  u32 bw[16], o[16];
  ld8(bases + 16 * w, bw);
  ld8(bases + 16 * w + 8, bw + 8);
  for (int i = 0; i < 16; i++) o[i] = 0;
  if (d != 0 && !affine_words_is_inf(bw)) {
    Affine b = affine_load_mont(bw);
    Xyzz acc = xyzz_inf();
    for (int bit = 7; bit >= 0; bit--) {
      acc = xyzz_dbl(acc);
      if ((d >> bit) & 1) acc = xyzz_madd(acc, b);
    }
    Affine a;
    if (xyzz_to_affine(acc, &a)) affine_store_mont(a, o);
  }
  st8(table + 16 * t, o);
  st8(table + 16 * t + 8, o + 8);
}
// table16[w][d] = d * 2^(16 w) * g1 for d < 65536 from the 8-bit table: T8[2w][d & 255] + T8[2w+1][d >> 8], written
// as XYZZ records (converted to affine in one batched pass afterwards).  Halves the additions per SRS power.
__global__ __launch_bounds__(128) void k_fb_table16(const u32* __restrict__ table8, u32* __restrict__ out_xyzz) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)16 << 16) return;
  const int w = (int)(t >> 16);
  const u32 d = (u32)t & 65535u, lo = d & 255u, hi = d >> 8;
  u32 a[16], b[16];
  ld8(table8 + 16 * ((size_t)(2 * w) * 256 + lo), a);
  ld8(table8 + 16 * ((size_t)(2 * w) * 256 + lo) + 8, a + 8);
  ld8(table8 + 16 * ((size_t)(2 * w + 1) * 256 + hi), b);
  ld8(table8 + 16 * ((size_t)(2 * w + 1) * 256 + hi) + 8, b + 8);
  Xyzz acc = xyzz_inf();
  if (!affine_words_is_inf(a)) acc = xyzz_from_affine(affine_load_mont(a));
  if (!affine_words_is_inf(b)) acc = xyzz_madd(acc, affine_load_mont(b));
  u32 o[32];
  xyzz_store(acc, o);
#pragma unroll
  for (int q = 0; q < 4; q++) st8(out_xyzz + 32 * t + 8 * q, o + 8 * q);
}
// atab[l][j] = alpha^(j * 2^(11 l)), l < 3, j < 2048, canonical Montgomery: alpha^e for e < 2^33 is then two
// products instead of a ~96-product square-and-multiply per SRS power.
constexpr int ATAB_BITS = 11;
__global__ void k_alpha_table(Words8k alpha_plain, u32* __restrict__ atab) {
  typedef FrParams R;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 << ATAB_BITS) return;
  const int l = t >> ATAB_BITS, j = t & ((1 << ATAB_BITS) - 1);
  Fe<R> a = fe_to_mont<R>(fe_unpack<R>(alpha_plain.w));
  Fe<R> v = fe_reduce<R>(fe_pow_u64<R>(a, (u64)j << (ATAB_BITS * l)));
  u32 o[8];
  fe_pack<R>(v, o);
  st8(atab + 8 * t, o);
}
// acc[i] = alpha^(first+i) * g1 as XYZZ (32 words); the affine conversion is batched afterwards.
template <int WB>   // window bits of the fixed-base table: 8 (32 windows, 512 KiB) or 16 (16 windows, 64 MiB)
__global__ __launch_bounds__(128) void k_fb_powers(Words8k alpha_plain, const u32* __restrict__ atab, const u32* __restrict__ table,
                                                   size_t first, size_t count, u32* __restrict__ out_xyzz) {
  typedef FrParams R;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const u64 e = (u64)(first + i);
  Fe<R> km;
  if (e >> (3 * ATAB_BITS)) {
    km = fe_pow_u64<R>(fe_to_mont<R>(fe_unpack<R>(alpha_plain.w)), e);
  } else {
    const u32 m = (1u << ATAB_BITS) - 1;
    u32 w0[8], w1[8], w2[8];
    ld8(atab + 8 * (size_t)(e & m), w0);
    ld8(atab + 8 * (size_t)((1u << ATAB_BITS) + ((e >> ATAB_BITS) & m)), w1);
    ld8(atab + 8 * (size_t)((2u << ATAB_BITS) + ((e >> (2 * ATAB_BITS)) & m)), w2);
    km = fe_mul<R>(fe_mul<R>(fe_unpack<R>(w0), fe_unpack<R>(w1)), fe_unpack<R>(w2));
  }
  Fe<R> k = fe_from_mont<R>(km);   // canonical alpha^(first+i)   (kzg.rs:33-36)
  u32 kw[8];
  fe_pack<R>(k, kw);
  Xyzz acc = xyzz_inf();
#pragma unroll 1
  for (int w = 0; w < 256 / WB; w++) {
    const u32 d = (WB == 8) ? ((kw[w >> 2] >> (8 * (w & 3))) & 255u) : ((kw[w >> 1] >> (16 * (w & 1))) & 65535u);
    if (d == 0) continue;
    u32 tw[16];
    const size_t idx = ((size_t)w << WB) + d;
    ld8(table + 16 * idx, tw);
    ld8(table + 16 * idx + 8, tw + 8);
    if (affine_words_is_inf(tw)) continue;
    acc = xyzz_madd_with<FeAsm>(acc, affine_load_mont(tw));
  }
  u32 o[32];
  xyzz_store(acc, o);
#pragma unroll
  for (int q = 0; q < 4; q++) st8(out_xyzz + 32 * i + 8 * q, o + 8 * q);
}

// XYZZ -> affine for a whole array with Montgomery's batch-inversion trick: one lane owns BATCH consecutive
// points, multiplies their denominators ZZ*ZZZ into running prefix products (parked in `scratch`, 9 words
// each), inverts the total ONCE (Fermat: all lanes in lockstep) and unwinds.  ~33 products per point instead
// of ~390.  Infinity (all-zero record) contributes the factor 1 and comes out as the all-zero encoding.
constexpr int BATCH_INV = 16;
__global__ __launch_bounds__(128) void k_xyzz_batch_to_affine(const u32* __restrict__ xyzz, size_t count, u32* __restrict__ scratch,
                                                              u32* __restrict__ out, int out_mont) {
  typedef FqParams P;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i0 = t * BATCH_INV;
  if (i0 >= count) return;
  const int len = (int)((count - i0 < (size_t)BATCH_INV) ? (count - i0) : (size_t)BATCH_INV);
  Fq run = fe_one<P>();
  for (int j = 0; j < len; j++) {
    u32 w[16];
    ld8(xyzz + 32 * (i0 + j) + 16, w);       // ZZ
    ld8(xyzz + 32 * (i0 + j) + 24, w + 8);   // ZZZ
    const bool inf = affine_words_is_inf(w);
    Fq d = FeAsm<P>::mul(fe_unpack<P>(w), fe_unpack<P>(w + 8));
    if (inf) d = fe_one<P>();
    // prefix BEFORE including element j
#pragma unroll
    for (int k = 0; k < P::L; k++) scratch[(i0 + j) * P::L + k] = run.l[k];
    run = FeAsm<P>::mul(run, d);
  }
  Fq inv = fe_inv<P>(run);
  for (int j = len - 1; j >= 0; j--) {
    u32 w[32];
#pragma unroll
    for (int q = 0; q < 4; q++) ld8(xyzz + 32 * (i0 + j) + 8 * q, w + 8 * q);
    const bool inf = affine_words_is_inf(w + 16);
    const Fq X = fe_unpack<P>(w), Y = fe_unpack<P>(w + 8), ZZ = fe_unpack<P>(w + 16), ZZZ = fe_unpack<P>(w + 24);
    Fq pre;
#pragma unroll
    for (int k = 0; k < P::L; k++) pre.l[k] = scratch[(i0 + j) * P::L + k];
    const Fq dinv = FeAsm<P>::mul(inv, pre);                       // 1 / (ZZ ZZZ) of element j
    Fq d = FeAsm<P>::mul(ZZ, ZZZ);
    if (inf) d = fe_one<P>();
    inv = FeAsm<P>::mul(inv, d);                                   // drop element j from the running inverse
    u32 o[16];
    if (inf) {
#pragma unroll
      for (int k = 0; k < 16; k++) o[k] = 0;
    } else {
      Affine a;
      a.x = fe_reduce<P>(FeAsm<P>::mul(X, FeAsm<P>::mul(dinv, ZZZ)));   // X / ZZ
      a.y = fe_reduce<P>(FeAsm<P>::mul(Y, FeAsm<P>::mul(dinv, ZZ)));    // Y / ZZZ
      if (out_mont) affine_store_mont(a, o); else affine_store_plain(a, o);
    }
    st8(out + 16 * (i0 + j), o);
    st8(out + 16 * (i0 + j) + 8, o + 8);
  }
}
int xyzz_batch_to_affine(const void* d_xyzz, size_t count, void* d_out, bool out_mont, hipStream_t s) {
  if (count == 0) return MZK_OK;
  u32* scratch;
  MZK_TRY(ws_get(WS_BATCHINV, count * FqParams::L * sizeof(u32), (void**)&scratch));
  const size_t threads = (count + BATCH_INV - 1) / BATCH_INV;
  hipLaunchKernelGGL(k_xyzz_batch_to_affine, dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, s, (const u32*)d_xyzz, count, scratch,
                     (u32*)d_out, out_mont ? 1 : 0);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

int kzg_setup_g1_dev(const uint64_t* alpha_host, const uint64_t* g1_host, size_t first, size_t count, void* d_powers_xy, hipStream_t s) {
  if (!alpha_host || !g1_host || (!d_powers_xy && count)) { set_error("kzg_setup: null pointer"); return MZK_E_ARG; }
  if (count == 0) return MZK_OK;
  if (!h_is_canonical(host_field(MZK_FIELD_FR), alpha_host) || !h_is_canonical(host_field(MZK_FIELD_FQ), g1_host) ||
      !h_is_canonical(host_field(MZK_FIELD_FQ), g1_host + 4)) { set_error("kzg_setup: operand not canonical"); return MZK_E_RANGE; }
  Words8k aw;
  Words16k gw;
  for (int i = 0; i < 4; i++) { aw.w[2 * i] = (u32)alpha_host[i]; aw.w[2 * i + 1] = (u32)(alpha_host[i] >> 32); }
  for (int i = 0; i < 8; i++) { gw.w[2 * i] = (u32)g1_host[i]; gw.w[2 * i + 1] = (u32)(g1_host[i] >> 32); }
  u32 *bases, *table, *acc, *atab;
  MZK_TRY(ws_get(WS_MISC_C, (size_t)(3 << ATAB_BITS) * 32, (void**)&atab));
  MZK_TRY(ws_get(WS_FB_TABLE8, 32 * 64 + 32 * 256 * 64, (void**)&bases));
  table = bases + 32 * 16;
  const bool wide = count >= ((size_t)1 << 16);      // the 64 MiB table pays for itself from ~2^16 powers on
  const size_t t16 = (size_t)16 << 16;
  MZK_TRY(ws_get(WS_XYZZ_TMP, (wide && t16 > count ? t16 : count) * 128, (void**)&acc));
  u32* table16 = nullptr;
  if (wide) MZK_TRY(ws_get(WS_FB_TABLE16, t16 * 64, (void**)&table16));
  // The fixed-base tables depend only on g1 (in practice always BN128::generator_g1()) and their construction is a
  // latency chain (248 serial doublings for the window bases + two inversions: ~1.4 ms, more than a whole 2^16-power
  // setup), so they are kept across calls, keyed by g1 and by the workspace generation.
  auto& cache = g_fb_cache[ctx().index];
  if (cache.gen != ws_generation() || memcmp(cache.g, g1_host, sizeof cache.g) != 0) {
    cache.gen = ws_generation(); cache.have8 = cache.have16 = false;
    memcpy(cache.g, g1_host, sizeof cache.g);
  }
  if (!cache.ready) MZK_HIP(hipEventCreateWithFlags(&cache.ready, hipEventDisableTiming));
  bool built = false;
  if (!cache.have8) {
    hipLaunchKernelGGL(k_fb_bases, dim3(1), dim3(32), 0, s, gw, bases);
    hipLaunchKernelGGL(k_fb_table, dim3(32), dim3(256), 0, s, (const u32*)bases, table);
    cache.have8 = true; built = true;
  }
  if (wide && !cache.have16) {
    hipLaunchKernelGGL(k_fb_table16, dim3((unsigned)(t16 / 128)), dim3(128), 0, s, (const u32*)table, acc);
    MZK_TRY(xyzz_batch_to_affine(acc, t16, table16, true, s));
    cache.have16 = true; built = true;
  }
  if (built) MZK_HIP(hipEventRecord(cache.ready, s));
  else MZK_HIP(hipStreamWaitEvent(s, cache.ready, 0));     // tables may have been built on another stream
  hipLaunchKernelGGL(k_alpha_table, dim3((3 << ATAB_BITS) / 256), dim3(256), 0, s, aw, atab);
  if (wide) {
    hipLaunchKernelGGL((k_fb_powers<16>), dim3((unsigned)((count + 127) / 128)), dim3(128), 0, s, aw, (const u32*)atab, (const u32*)table16,
                       first, count, acc);
  } else {
    hipLaunchKernelGGL((k_fb_powers<8>), dim3((unsigned)((count + 127) / 128)), dim3(128), 0, s, aw, (const u32*)atab, (const u32*)table,
                       first, count, acc);
  }
  MZK_HIP(hipGetLastError());
  return xyzz_batch_to_affine(acc, count, d_powers_xy, false, s);
}

// ---- open: suffix Horner b_i = c_i + u b_{i+1} -----------------------------------------------------------
constexpr int OPEN_K_LOG = 5;   // chunk = 32: the per-lane recurrence is a serial chain, so short chunks (more levels, all tiny) win
constexpr size_t OPEN_K = (size_t)1 << OPEN_K_LOG;
typedef Fe<FrParams> FrE;
__device__ __forceinline__ FrE fr_gload(const u32* __restrict__ g, size_t i) {
  u32 w[8];
  ld8(g + 8 * i, w);
  return fe_unpack<FrParams>(w);
}
__device__ __forceinline__ void fr_gstore(u32* __restrict__ g, size_t i, const FrE& v) {
  u32 w[8];
  fe_pack<FrParams>(v, w);
  st8(g + 8 * i, w);
}
// h[m] = sum_{t in chunk m} c[t] u^(t - m K)
__global__ __launch_bounds__(64) void k_open_chunk_eval(const u32* __restrict__ c, size_t n, Words8k u_mont, u32* __restrict__ h) {
  const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = m << OPEN_K_LOG;
  if (lo >= n) return;
  const size_t hi = (lo + OPEN_K < n) ? lo + OPEN_K : n;
  const FrE u = fe_unpack<FrParams>(u_mont.w);
  FrE acc = fr_gload(c, hi - 1);
  for (size_t t = hi - 1; t-- > lo;) acc = fe_add<FrParams>(fe_mul<FrParams>(acc, u), fr_gload(c, t));
  fr_gstore(h, m, fe_reduce<FrParams>(acc));
}
// b[t] = c[t] + u b[t+1] inside chunk m, with b[(m+1) K] = carry[m+1] (0 past the end)
__global__ __launch_bounds__(64) void k_open_chunk_fill(const u32* __restrict__ c, size_t n, Words8k u_mont, const u32* __restrict__ carry,
                                                         size_t ncarry, u32* __restrict__ b) {
  const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = m << OPEN_K_LOG;
  if (lo >= n) return;
  const size_t hi = (lo + OPEN_K < n) ? lo + OPEN_K : n;
  const FrE u = fe_unpack<FrParams>(u_mont.w);
  FrE acc = (carry != nullptr && m + 1 < ncarry) ? fr_gload(carry, m + 1) : fe_zero<FrParams>();
  for (size_t t = hi; t-- > lo;) {
    acc = fe_reduce<FrParams>(fe_add<FrParams>(fe_mul<FrParams>(acc, u), fr_gload(c, t)));
    fr_gstore(b, t, acc);
  }
}

// One synthetic division of the level-0 array `c` (n elements) by (X - u): fills b (n elements) with
// b_i = c_i + u b_{i+1}.  Scratch: hbuf / bup hold the upper levels.
static int suffix_horner(const u32* c, size_t n, const uint64_t* u_host, u32* b, u32* hbuf, u32* bup, hipStream_t s) {
  const HostField* fr = host_field(MZK_FIELD_FR);
  size_t lens[8];
  int nlev = 0;
  lens[0] = n;
  while (lens[nlev] > OPEN_K) { lens[nlev + 1] = (lens[nlev] + OPEN_K - 1) >> OPEN_K_LOG; nlev++; }
  Words8k um[8];
  {
    uint64_t ul[4] = {u_host[0], u_host[1], u_host[2], u_host[3]};
    uint64_t two[4] = {2, 0, 0, 0}, rmod[4], t[4];
    h_powmod_u64(fr, rmod, two, 261);
    for (int l = 0; l <= nlev; l++) {
      h_mulmod(fr, t, ul, rmod);
      for (int i = 0; i < 4; i++) { um[l].w[2 * i] = (u32)t[i]; um[l].w[2 * i + 1] = (u32)(t[i] >> 32); }
      h_powmod_u64(fr, ul, ul, OPEN_K);
    }
  }
  const u32* level_in[8];
  u32* level_b[8];
  level_in[0] = c;
  level_b[0] = b;
  size_t off = 0;
  for (int l = 1; l <= nlev; l++) {
    level_in[l] = hbuf + off * 8;
    level_b[l] = bup + off * 8;
    off += lens[l];
  }
  for (int l = 0; l < nlev; l++) {
    const size_t chunks = lens[l + 1];
    hipLaunchKernelGGL(k_open_chunk_eval, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, s, level_in[l], lens[l], um[l], (u32*)level_in[l + 1]);
  }
  for (int l = nlev; l >= 0; l--) {
    const size_t chunks = (lens[l] + OPEN_K - 1) >> OPEN_K_LOG;
    const u32* carry = (l == nlev) ? nullptr : level_b[l + 1];
    hipLaunchKernelGGL(k_open_chunk_fill, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, s, level_in[l], lens[l], um[l], carry,
                       (l == nlev) ? (size_t)0 : lens[l + 1], level_b[l]);
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// batch_open_kzg (kzg.rs:74-88).  y_i = f(u_i) is b_0 of the synthetic division of f by (X - u_i).  The
// quotient of f by prod (X - u_i) -- which equals (f - I)/Z because I = f mod Z -- is k successive synthetic
// divisions (each drops the remainder b_0).  d_ys: k * 8 words, d_w_xy: 16 words.
int kzg_batch_open_dev(const void* d_coef, size_t n, const uint64_t* us_host, size_t k, const void* d_points, int point_kind,
                       size_t table_stride, void* d_ys, void* d_w_xy, hipStream_t s) {
  if (!d_w_xy || (!d_coef && n) || ((!us_host || !d_ys) && k)) { set_error("batch_open: null pointer"); return MZK_E_ARG; }
  const HostField* fr = host_field(MZK_FIELD_FR);
  for (size_t i = 0; i < k; i++) if (!h_is_canonical(fr, us_host + 4 * i)) { set_error("batch_open: u not canonical"); return MZK_E_RANGE; }
  if (n == 0) {  // zero polynomial: every y = 0, quotient empty
    if (k) MZK_HIP(hipMemsetAsync(d_ys, 0, k * 32, s));
    MZK_HIP(hipMemsetAsync(d_w_xy, 0, 64, s));
    return MZK_OK;
  }
  size_t total_up = 0;
  for (size_t l = (n + OPEN_K - 1) >> OPEN_K_LOG; ; l = (l + OPEN_K - 1) >> OPEN_K_LOG) { total_up += l; if (l <= 1) break; }
  u32 *bA, *bB, *hbuf, *bup;
  MZK_TRY(ws_get(WS_MISC_A, n * 32, (void**)&bA));
  MZK_TRY(ws_get(WS_MISC_D, n * 32, (void**)&bB));
  MZK_TRY(ws_get(WS_MISC_B, (total_up + 2) * 32, (void**)&hbuf));
  MZK_TRY(ws_get(WS_MISC_C, (total_up + 2) * 32, (void**)&bup));
  // evaluations of the ORIGINAL f
  for (size_t i = 0; i < k; i++) {
    MZK_TRY(suffix_horner((const u32*)d_coef, n, us_host + 4 * i, bA, hbuf, bup, s));
    MZK_HIP(hipMemcpyAsync((char*)d_ys + 32 * i, bA, 32, hipMemcpyDeviceToDevice, s));
  }
  // successive quotients
  const u32* cur = (const u32*)d_coef;
  size_t len = n;
  u32* dst = bA;
  for (size_t i = 0; i < k && len > 0; i++) {
    MZK_TRY(suffix_horner(cur, len, us_host + 4 * i, dst, hbuf, bup, s));
    cur = dst + 8;          // q_j = b_{j+1}
    len -= 1;
    dst = (dst == bA) ? bB : bA;
  }
  if (k > n) len = 0;
  return msm_dev_impl(cur, d_points, len, point_kind, table_stride, d_w_xy, false, s);
}

// d_y: 8 words; d_w_xy: 16 words.
int kzg_open_dev(const void* d_coef, size_t n, const uint64_t* u_host, const void* d_points, int point_kind, size_t table_stride,
                 void* d_y, void* d_w_xy, void* d_q_out, hipStream_t s, bool value_only) {
  if (!u_host || !d_y || (!d_w_xy && !d_q_out && !value_only) || (!d_coef && n) || (!d_points && n > 1 && !d_q_out && !value_only)) { set_error("kzg_open: null pointer"); return MZK_E_ARG; }
  const HostField* fr = host_field(MZK_FIELD_FR);
  if (!h_is_canonical(fr, u_host)) { set_error("kzg_open: u not canonical"); return MZK_E_RANGE; }
  if (n == 0) {  // empty polynomial: y = 0, quotient empty -> infinity
    MZK_HIP(hipMemsetAsync(d_y, 0, 32, s));
    if (d_w_xy) MZK_HIP(hipMemsetAsync(d_w_xy, 0, 64, s));
    return MZK_OK;
  }
  // level arrays: L0 = coef (n), L1 = chunk values (ceil(n/K)), ...
  size_t lens[8];
  int nlev = 0;
  lens[0] = n;
  while (lens[nlev] > OPEN_K) { lens[nlev + 1] = (lens[nlev] + OPEN_K - 1) >> OPEN_K_LOG; nlev++; }
  size_t total_up = 0;
  for (int l = 1; l <= nlev; l++) total_up += lens[l];
  u32 *bbuf, *hbuf, *bup;
  MZK_TRY(ws_get(WS_MISC_A, n * 32, (void**)&bbuf));                  // b of level 0
  MZK_TRY(ws_get(WS_MISC_B, (total_up + 1) * 32, (void**)&hbuf));     // h of levels 1..nlev
  MZK_TRY(ws_get(WS_MISC_C, (total_up + 1) * 32, (void**)&bup));      // b of levels 1..nlev
  // u^(K^l) in Montgomery form, on the host (parameter math)
  Words8k um[8];
  {
    uint64_t ul[4] = {u_host[0], u_host[1], u_host[2], u_host[3]};
    uint64_t two[4] = {2, 0, 0, 0}, rmod[4], t[4];
    h_powmod_u64(fr, rmod, two, 261);
    for (int l = 0; l <= nlev; l++) {
      h_mulmod(fr, t, ul, rmod);
      for (int i = 0; i < 4; i++) { um[l].w[2 * i] = (u32)t[i]; um[l].w[2 * i + 1] = (u32)(t[i] >> 32); }
      h_powmod_u64(fr, ul, ul, OPEN_K);
    }
  }
  const u32* level_in[8];
  u32* level_b[8];
  level_in[0] = (const u32*)d_coef;
  level_b[0] = bbuf;
  size_t off = 0;
  for (int l = 1; l <= nlev; l++) {
    level_in[l] = hbuf + off * 8;
    level_b[l] = bup + off * 8;
    off += lens[l];
  }
  for (int l = 0; l < nlev; l++) {
    const size_t chunks = lens[l + 1];
    hipLaunchKernelGGL(k_open_chunk_eval, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, s, level_in[l], lens[l], um[l],
                       (u32*)level_in[l + 1]);
  }
  for (int l = nlev; l >= 0; l--) {
    const size_t chunks = (lens[l] + OPEN_K - 1) >> OPEN_K_LOG;
    const u32* carry = (l == nlev) ? nullptr : level_b[l + 1];
    hipLaunchKernelGGL(k_open_chunk_fill, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, s, level_in[l], lens[l], um[l], carry,
                       (l == nlev) ? (size_t)0 : lens[l + 1], level_b[l]);
  }
  MZK_HIP(hipGetLastError());
  MZK_HIP(hipMemcpyAsync(d_y, bbuf, 32, hipMemcpyDeviceToDevice, s));   // y = b_0
  if (value_only) return MZK_OK;          // f(u) alone (the sharded opening's first pass: the slice's value)
  if (d_q_out) {   // quotient only (sharded opening: every rank MSMs its own slice of q)
    if (n > 1) MZK_HIP(hipMemcpyAsync(d_q_out, bbuf + 8, (n - 1) * 32, hipMemcpyDeviceToDevice, s));
    return MZK_OK;
  }
  // w = MSM(q, powers), q_j = b_{j+1}, j < n - 1     (kzg.rs:70)
  return msm_dev_impl(bbuf + 8, d_points, n - 1, point_kind, table_stride, d_w_xy, false, s);
}

// ---- open_kzg of MANY short polynomials, polynomial j at its own point u_j (kzg.rs:61-72 once per polynomial, as
// das/avail.rs:132 does per cell): ONE workgroup per polynomial solves the suffix recurrence b_i = c_i + u b_{i+1} --
// lane L owns the K = ceil(n / 256) coefficients [L K, (L + 1) K): chunk value at u by Horner, a log-step suffix scan of
// the 256 chunk values with u^K, u^2K, ... through LDS, then the chunk is filled from its successor's carry.  y_j = b_0,
// q_j = b_1 .. b_{n-1} (row j of `q`, rows q_stride_words apart); the witness MSMs then run as one grid-batched pass
// (msm_many_dev_impl).  Field arithmetic is exact, so the canonical outputs equal the chunked single-opening path bit for bit.
constexpr int OPENM_THREADS = 256;
constexpr size_t OPENM_MAX_N = (size_t)1 << 14;
constexpr int PUT_WORDS_MAX = 112;                 // 32-byte values per launch through the kernel-argument buffer (3.5 KiB)
struct PutBatch { u32 w[PUT_WORDS_MAX][8]; };
__global__ void k_put_words8(PutBatch b, int count, u32* __restrict__ dst) {
  const int i = threadIdx.x;
  if (i < count) st8(dst + 8 * i, b.w[i]);
}
__global__ __launch_bounds__(OPENM_THREADS) void k_open_many(const u32* __restrict__ coefs, size_t n, size_t stride_words, const u32* __restrict__ us,
                                                              u32* __restrict__ q, size_t q_stride_words, u32* __restrict__ ys) {
  typedef FrParams P;
  __shared__ u32 S[OPENM_THREADS][P::L];
  const size_t j = blockIdx.x;
  const int L = threadIdx.x;
  const u32* c = coefs + j * stride_words;
  u32 uw[8];
  ld8(us + 8 * j, uw);
  const FrE u = fe_to_mont<P>(fe_unpack<P>(uw));            // u R: fe_mul(x, u) = x u for plain x
  const size_t K = (n + OPENM_THREADS - 1) / OPENM_THREADS;
  const size_t lo = (size_t)L * K;
  const size_t hi = (lo + K < n) ? lo + K : n;
  FrE mine = fe_zero<P>();
  if (lo < n) {
    mine = fr_gload(c, hi - 1);
    for (size_t t = hi - 1; t-- > lo;) mine = fe_add<P>(fe_mul<P>(mine, u), fr_gload(c, t));
    mine = fe_reduce<P>(mine);
  }
  FrE pw = fe_one<P>(), base = u;                           // u^K in Montgomery form
  for (size_t k = K; k; k >>= 1) {
    if (k & 1) pw = fe_mul<P>(pw, base);
    base = fe_sqr<P>(base);
  }
#pragma unroll
  for (int i = 0; i < P::L; i++) S[L][i] = mine.l[i];
  for (int d = 1; d < OPENM_THREADS; d <<= 1) {
    __syncthreads();
    FrE other = fe_zero<P>();
    if (L + d < OPENM_THREADS) {
#pragma unroll
      for (int i = 0; i < P::L; i++) other.l[i] = S[L + d][i];
    }
    __syncthreads();
    mine = fe_reduce<P>(fe_add<P>(mine, fe_mul<P>(other, pw)));
#pragma unroll
    for (int i = 0; i < P::L; i++) S[L][i] = mine.l[i];
    pw = fe_sqr<P>(pw);
  }
  __syncthreads();
  if (lo >= n) return;
  FrE acc = fe_zero<P>();                                   // b at index (L + 1) K
  if (L + 1 < OPENM_THREADS) {
#pragma unroll
    for (int i = 0; i < P::L; i++) acc.l[i] = S[L + 1][i];
  }
  u32* qj = q + j * q_stride_words;
  for (size_t t = hi; t-- > lo;) {
    acc = fe_reduce<P>(fe_add<P>(fe_mul<P>(acc, u), fr_gload(c, t)));
    if (t > 0) fr_gstore(qj, t - 1, acc);
    else fr_gstore(ys, j, acc);
  }
}

bool kzg_open_many_supported(const mzk_srs* srs, size_t n) { return n <= OPENM_MAX_N && srs_many_capable(srs); }
// d_ys: count * 8 words, d_ws_xy: count * 16 words.
int kzg_open_many_dev(const mzk_srs* srs, const void* d_coefs, size_t n, size_t count, const uint64_t* us_host, void* d_ys, void* d_ws_xy, hipStream_t s) {
  if (count == 0) return MZK_OK;
  if (!srs || !us_host || !d_ys || !d_ws_xy || (!d_coefs && n)) { set_error("kzg_open_many: null pointer"); return MZK_E_ARG; }
  const HostField* fr = host_field(MZK_FIELD_FR);
  for (size_t i = 0; i < count; i++) if (!h_is_canonical(fr, us_host + 4 * i)) { set_error("kzg_open: u not canonical"); return MZK_E_RANGE; }
  if (n == 0) {  // empty polynomials: y = 0, quotient empty -> infinity
    MZK_HIP(hipMemsetAsync(d_ys, 0, count * 32, s));
    MZK_HIP(hipMemsetAsync(d_ws_xy, 0, count * 64, s));
    return MZK_OK;
  }
  size_t per_pass = (((size_t)1 << 22) + n - 1) / n;
  u32 *d_us, *d_q;
  const size_t first_cnt = count < per_pass ? count : per_pass;
  MZK_TRY(ws_get(WS_MISC_B, first_cnt * 32, (void**)&d_us));
  MZK_TRY(ws_get(WS_MISC_A, first_cnt * n * 32, (void**)&d_q));
  for (size_t first = 0; first < count; first += per_pass) {
    const size_t cnt = (count - first < per_pass) ? count - first : per_pass;
    for (size_t k = 0; k < cnt; k += PUT_WORDS_MAX) {
      PutBatch b;
      const int m = (int)((cnt - k < (size_t)PUT_WORDS_MAX) ? cnt - k : (size_t)PUT_WORDS_MAX);
      for (int i = 0; i < m; i++)
        for (int l = 0; l < 4; l++) {
          const uint64_t v = us_host[4 * (first + k + i) + l];
          b.w[i][2 * l] = (u32)v; b.w[i][2 * l + 1] = (u32)(v >> 32);
        }
      hipLaunchKernelGGL(k_put_words8, dim3(1), dim3(128), 0, s, b, m, d_us + 8 * k);
    }
    hipLaunchKernelGGL(k_open_many, dim3((unsigned)cnt), dim3(OPENM_THREADS), 0, s, (const u32*)d_coefs + first * n * 8, n, n * 8, (const u32*)d_us, d_q, n * 8,
                       (u32*)d_ys + first * 8);
    MZK_HIP(hipGetLastError());
    // w_j = MSM(q_j, powers), q_j of n - 1 coefficients     (kzg.rs:70)
    MZK_TRY(msm_many_srs(srs, d_q, n - 1, n, cnt, (u32*)d_ws_xy + first * 16, s));
  }
  return MZK_OK;
}

}  // namespace mzk
