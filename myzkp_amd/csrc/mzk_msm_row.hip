// mzk_msm_row.hip -- the latency-bound tails of the MSM on ROW-cooperative group operations (mzk_row.h: one point operation
// per wave, a field element spread over a DPP row): the late halving steps of the bucket reduction, its weighted sum and
// tree, the window Horner of the generic layout, and the multi-GPU fold of XYZZ partials.  Stands behind the same reference
// code as mzk_msm.hip (Polynomial::eval_with_powers_on_curve, polynomial.rs:156-165); the early, wide halving steps and
// everything throughput-bound stay in mzk_msm.hip.  An addition here is ~1.3 us on an idle GPU against ~3.6 us for the DPP-quad
// form (mzk_coop.h), a doubling ~1.0 against ~2.9 -- and these chains are 30 (merged layout) to 150 (generic) operations long.
#include "mzk_common.h"
#include "mzk_ec.h"
#include "mzk_coop.h"
#include "mzk_row.h"
#include "mzk_inv_wave.h"
#include "mzk_affine_wave.h"

namespace mzk {

using rowop::Lane;
using rowop::Pt;

__device__ __forceinline__ void halve_indices(int lgB, int t, size_t id, size_t* lo, size_t* hi) {
  const int lgh = lgB - t - 1;
  const size_t a = id >> lgh, j = id & (((size_t)1 << lgh) - 1);
  *lo = ((a == 0) ? 0 : ((size_t)1 << (lgB - a))) + j;
  *hi = *lo + ((size_t)1 << lgh);
}
__device__ __noinline__ void halve_op_quad(u32* __restrict__ buf, int lgB, int t, size_t id, int lane) {
  size_t lo, hi;
  halve_indices(lgB, t, id, &lo, &hi);
  const Xyzz x = xyzz_gload_quad(buf, lo, lane), y = xyzz_gload_quad(buf, hi, lane);
  xyzz_gstore_quad(buf, lo, xyzz_add_quad(x, y, lane), lane);
}

// -DMZK_TAIL_TRACE (what-if builds only, tools/timing/tail_trace.py): 100-MHz timestamps of the tail's phases
#ifdef MZK_TAIL_TRACE
__device__ unsigned long long g_tail_trace[64];
#define MZK_TT(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { g_tail_trace[(i)] = wall_clock64(); if ((i) == 0 || (i) == 60) g_tail_trace[(i) + 1 + ((i) == 0 ? 61 : 0)] = clock64(); } } while (0)
extern "C" int mzk_debug_tail_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_trace), sizeof(g_tail_trace)) == hipSuccess ? 0 : -1;
}
#else
#define MZK_TT(i)
#endif

constexpr int RTAIL_THREADS = 256;               // four waves = one per SIMD: see below
constexpr int RTAIL_WAVES = RTAIL_THREADS / 64;
constexpr int RTAIL_QUADS = RTAIL_THREADS / 4;
constexpr int RTAIL_LDS_SLOTS = 512;             // live entries the workgroup keeps in LDS (64 KiB) once they fit
constexpr int RTAIL_GROUPS = 4;                  // Horner groups of the weighted sum: one wave per SIMD
struct TailPlan { int b[RTAIL_GROUPS + 1]; };    // wave g sums the terms q in [b[g], b[g + 1])

// ONE copy of the group operations that the tail runs once here and once there: what such code waits for is its own
// instruction fetch as much as the arithmetic (the fully inlined kernel was 142 KB against a 64-KB instruction cache;
// tools/timing/tail_trace.py).  Loops keep their operations inlined: a call costs ~0.25 us of register saves and restores.
__device__ __noinline__ Pt row_add_shared(Pt a, Pt b, Lane ln) { return rowop::add(a, b, ln); }
// buf[lo] += buf[hi] by a DPP quad; buf in LDS or global memory
__device__ __noinline__ void quad_add_lds(int lo, int hi, int ql) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const Xyzz x = xyzz_gload_quad(lds, (size_t)lo, ql), y = xyzz_gload_quad(lds, (size_t)hi, ql);
  xyzz_gstore_quad(lds, (size_t)lo, xyzz_add_quad(x, y, ql), ql);
}

// Remaining steps t_start .. lgB-1 of a bucket set inside one workgroup, then  buf[0] + sum_q 2^q buf[2^q]  and, for a single
// bucket set, the affine conversion (one safegcd inversion).
// One CU's issue rate bounds this kernel, not only the chain (tools/timing/tail_trace.py: a row operation keeps its SIMD
// ~85 % busy -- four waves of a SIMD doubling at once took 1.9 us per doubling each, not 1.0 -- and 1024 lanes of quad
// additions were capped at 128 VGPRs and spilled).  Hence
//  * four waves, one per SIMD, 256 VGPRs each; the host hands over at steps of <= 64 additions (one round of 64 quads);
//  * the live entries move into LDS as soon as they fit (no global round trip per step);
//  * a step runs on row operations only when it has at most `row_max` (<= 4) additions, on DPP quads above;
//  * the weighted sum is FOUR Horner chains (terms split by the host so that the chains are equally long: max(G) doublings +
//    |G| - 1 additions each) and a two-level tree -- 44 doublings for 16 terms where "wave q doubles q times" needed 120.
__global__ __launch_bounds__(RTAIL_THREADS) void k_reduce_tail_row(u32* __restrict__ buckets, int lgB, int t_start, u32* __restrict__ out, int finish_affine,
                                                                   int row_max, TailPlan plan) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* res = lds + RTAIL_LDS_SLOTS * 32;
  u32* buf = buckets + ((size_t)blockIdx.x << lgB) * 32;
  const Lane ln = rowop::lane_init();
  const int wave = threadIdx.x >> 6;
  MZK_TT(0);
  int t = t_start;
  for (; t < lgB && ((size_t)(t + 1) << (lgB - t)) > (size_t)RTAIL_LDS_SLOTS; t++) {      // too many live entries for LDS: in place
    const size_t total = (size_t)(t + 1) << (lgB - t - 1);
    for (size_t id = threadIdx.x >> 2; id < total; id += RTAIL_QUADS) halve_op_quad(buf, lgB, t, id, (int)(threadIdx.x & 3));
    __syncthreads();
    MZK_TT(1 + t);
  }
  // live now: region 0 = [0, W) and regions a = 1 .. t0 = [2^(lgB - a), + W), W = 2^(lgB - t0); region a -> LDS slots [a W, + W).
  // The regions the later steps create (a > t0) lie inside region 0's slots, at their global index.
  const int t0 = t, lgW = lgB - t0;
  {
    const int n16 = ((t0 + 1) << lgW) * 8;
    for (int k = threadIdx.x; k < n16; k += RTAIL_THREADS) {
      const int e = k >> 3, a = e >> lgW, j = e & ((1 << lgW) - 1);
      const size_t src = (a ? ((size_t)1 << (lgB - a)) : 0) + (size_t)j;
      reinterpret_cast<uint4*>(lds)[k] = reinterpret_cast<const uint4*>(buf + src * 32)[k & 7];
    }
  }
  __syncthreads();
  MZK_TT(39);
  for (; t < lgB; t++) {
    const int lgh = lgB - t - 1;
    const int total = (t + 1) << lgh;
    if (total > row_max) {
      for (int id = threadIdx.x >> 2; id < total; id += RTAIL_QUADS) {
        const int a = id >> lgh, j = id & ((1 << lgh) - 1);
        const int lo = ((a == 0) ? 0 : (a > t0) ? (1 << (lgB - a)) : (a << lgW)) + j;
        quad_add_lds(lo, lo + (1 << lgh), (int)(threadIdx.x & 3));
      }
    } else {
      for (int id = wave; id < total; id += RTAIL_WAVES) {
        const int a = id >> lgh, j = id & ((1 << lgh) - 1);
        const int lo = ((a == 0) ? 0 : (a > t0) ? (1 << (lgB - a)) : (a << lgW)) + j;
        const Pt x = rowop::load(lds + lo * 32, ln), y = rowop::load(lds + (lo + (1 << lgh)) * 32, ln);
        rowop::store(lds + lo * 32, row_add_shared(x, y, ln), ln);
      }
    }
    __syncthreads();
    MZK_TT(1 + t);
  }
  {
    // term q = 2^q buf[2^q]; buf[2^q] is the start of region lgB - q
    auto term = [&](int q) { const int a = lgB - q; return rowop::load(lds + ((a > t0) ? (1 << q) : (a << lgW)) * 32, ln); };
    const int lo = plan.b[wave], hi = plan.b[wave + 1];
    // one chain: v = term(hi-1); then for k = hi-2 .. 0: v = 2 v (+ term(k) while k >= lo); wave 0 ends with + buf[0].
    // ONE inlined doubling and ONE inlined addition in a loop (hot after the first trip; a call costs ~0.25 us in saves / restores)
    Pt v = (hi > lo) ? term(hi - 1) : rowop::pt_inf();
#pragma unroll 1
    for (int k = (hi > lo) ? hi - 2 : -1; k >= -1; k--) {
      if (k >= 0) v = rowop::dbl(v, ln);
      if (k >= lo || (k < 0 && wave == 0)) v = rowop::add(v, (k >= 0) ? term(k) : rowop::load(lds, ln), ln);
    }
    rowop::store(res + wave * 32, v, ln);
  }
  __syncthreads();
  MZK_TT(40);
  if (wave < 2) {
    const Pt x = rowop::load(res + wave * 32, ln), y = rowop::load(res + (wave + 2) * 32, ln);
    rowop::store(res + wave * 32, row_add_shared(x, y, ln), ln);
  }
  __syncthreads();
  MZK_TT(41);
  if (wave != 0) return;
  {
    const Pt x = rowop::load(res, ln), y = rowop::load(res + 32, ln);
    rowop::store(res, row_add_shared(x, y, ln), ln);
  }
  MZK_TT(42);
  if (finish_affine) {          // one affine point per bucket set (a single set for one commit; one per polynomial for mzk_*_many)
    wave_store_affine(res, out + (size_t)blockIdx.x * 16);
    MZK_TT(60);
    return;
  }
  if (threadIdx.x < 32) out[(size_t)blockIdx.x * 32 + threadIdx.x] = res[threadIdx.x];
}

// total = sum_w 2^(c w) R_w (Horner over the bucket sets of the generic layout: c doublings per window, inherently serial)
// on ONE wave, then affine or the XYZZ partial record.
__global__ __launch_bounds__(64) void k_window_combine_row(const u32* __restrict__ wsum, int nwin, int c, int out_xyzz, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 sh[32];
  const Lane ln = rowop::lane_init();
  Pt tot = rowop::load(wsum + (size_t)(nwin - 1) * 32, ln);
  for (int win = nwin - 2; win >= 0; win--) {
    for (int d = 0; d < c; d++) tot = rowop::dbl(tot, ln);
    tot = rowop::add(tot, rowop::load(wsum + (size_t)win * 32, ln), ln);
  }
  if (out_xyzz) { rowop::store(out, tot, ln); return; }
  rowop::store(sh, tot, ln);
  __syncthreads();
  wave_store_affine(sh, out);
}
// fold `count` XYZZ partials (the gathered records of the ranks / contexts) into one affine point
// (grid: one wave per result -- workgroup g folds partials [g count, (g + 1) count) into out[g]; one workgroup for the ranks' fold)
__global__ __launch_bounds__(64) void k_fold_partials_row(const u32* __restrict__ partials, int count, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 sh[32];
  const Lane ln = rowop::lane_init();
  partials += (size_t)blockIdx.x * (size_t)count * 32;
  Pt tot = rowop::pt_inf();
  for (int i = 0; i < count; i++) tot = rowop::add(tot, rowop::load(partials + (size_t)i * 32, ln), ln);
  rowop::store(sh, tot, ln);
  __syncthreads();
  wave_store_affine(sh, out + (size_t)blockIdx.x * 16);
}

// Horner groups for lgB terms: contiguous ranges [b[g], b[g + 1]), g = 0 .. 3, minimising the longest chain --
// (top term) doublings + (terms - 1) additions, group 0 one addition more (the unweighted sum); 1.0 / 1.4 us each.
static TailPlan tail_plan(int lgB) {
  static TailPlan cache[32];
  static bool have[32];
  if (lgB < 0 || lgB >= 32) lgB = 31;
  if (have[lgB]) return cache[lgB];
  auto cost = [](int g, int lo, int hi) { return hi > lo ? 10 * (hi - 1) + 14 * (hi - lo - 1) + (g == 0 ? 14 : 0) : 0; };
  TailPlan best{};
  int best_max = 1 << 30, best_sum = 1 << 30;
  for (int b1 = 0; b1 <= lgB; b1++)
    for (int b2 = b1; b2 <= lgB; b2++)
      for (int b3 = b2; b3 <= lgB; b3++) {
        const int b[5] = {0, b1, b2, b3, lgB};
        int mx = 0, sum = 0;
        for (int g = 0; g < 4; g++) { const int c = cost(g, b[g], b[g + 1]); mx = c > mx ? c : mx; sum += c; }
        if (mx < best_max || (mx == best_max && sum < best_sum)) {
          best_max = mx; best_sum = sum;
          for (int g = 0; g < 5; g++) best.b[g] = b[g];
        }
      }
  cache[lgB] = best; have[lgB] = true;
  return best;
}
static int tail_row_max() {      // widest step that still runs as row operations (A/B: tools/timing/small_latency.py)
  static const int v = tune_int("MZK_TAIL_ROW_MAX", 4);
  return v;
}
int launch_reduce_tail_row(u32* buckets, int lgB, int t_start, int sets, u32* out, int finish_affine, hipStream_t s) {
  constexpr size_t LDS_BYTES = (size_t)(RTAIL_LDS_SLOTS + RTAIL_GROUPS) * 128;
  bool& configured = ctx().attr_done[ATTR_TAIL_ROW];
  if (!configured) {
    MZK_HIP(hipFuncSetAttribute((const void*)k_reduce_tail_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    configured = true;
  }
  hipLaunchKernelGGL(k_reduce_tail_row, dim3((unsigned)sets), dim3(RTAIL_THREADS), LDS_BYTES, s, buckets, lgB, t_start, out, finish_affine, tail_row_max(),
                     tail_plan(lgB));
  return MZK_OK;
}
int launch_window_combine_row(const u32* wsum, int nwin, int c, int out_xyzz, u32* out, hipStream_t s) {
  hipLaunchKernelGGL(k_window_combine_row, dim3(1), dim3(64), 0, s, wsum, nwin, c, out_xyzz, out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
int launch_fold_partials_row(const u32* partials, int count, u32* out, hipStream_t s) {
  hipLaunchKernelGGL(k_fold_partials_row, dim3(1), dim3(64), 0, s, partials, count, out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
// `groups` independent folds of `count` consecutive partials each -> groups affine points
int launch_fold_partial_groups_row(const u32* partials, int count, size_t groups, u32* out, hipStream_t s) {
  if (groups == 0) return MZK_OK;
  hipLaunchKernelGGL(k_fold_partials_row, dim3((unsigned)groups), dim3(64), 0, s, partials, count, out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

// ---- self-test: row operations against the plain formulas of mzk_ec.h -----------------------------------------------------
// Pair i: a = [k_i] of a synthetic point in XYZZ form with a random ZZ, b by case i % 8: an independent point (0..3), the same
// point in another representation (4: P + P), its negative (5: P + (-P)), infinity on either side (6, 7).  Row kernels write
// a + b and 2 a as packed records; the checker recomputes both with xyzz_add / xyzz_dbl on one lane and compares as group
// elements (cross-multiplied coordinates), every record also for the storage bound value < 2.5 p.
__device__ __forceinline__ u64 st_mix(u64& s) {
  u64 z = (s += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ Fq st_rand_fq(u64& s) {
  u32 w[8];
  for (int i = 0; i < 4; i++) { const u64 v = st_mix(s); w[2 * i] = (u32)v; w[2 * i + 1] = (u32)(v >> 32); }
  w[7] &= 0x0fffffffu;                                         // < 2^252 < p
  return fe_unpack<FqParams>(w);
}
__device__ __forceinline__ Xyzz st_rescale(const Affine& p, const Fq& z) {     // (x z^2, y z^3, z^2, z^3)
  typedef FqParams P;
  Xyzz r;
  const Fq z2 = fe_sqr<P>(z), z3 = fe_mul<P>(z2, z);
  r.X = fe_mul<P>(p.x, z2); r.Y = fe_mul<P>(p.y, z3); r.ZZ = z2; r.ZZZ = z3;
  return r;
}
__global__ __launch_bounds__(128) void k_rowtest_prepare(const u32* __restrict__ pts_mont, size_t n, u64 seed, u32* __restrict__ a_out, u32* __restrict__ b_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s = seed ^ (0x51ed27ULL * (i + 1));
  const Affine p = affine_load_mont(pts_mont + i * 16), q = affine_load_mont(pts_mont + ((i * 7 + 3) % n) * 16);
  Fq z1 = st_rand_fq(s), z2 = st_rand_fq(s);
  z1.l[0] |= 1; z2.l[0] |= 1;                                  // non-zero
  Xyzz a = st_rescale(p, z1), b;
  switch (i & 7) {
    case 4: b = st_rescale(p, z2); break;
    case 5: b = st_rescale(affine_neg(p), z2); break;
    case 6: b = xyzz_inf(); break;
    case 7: b = a; a = xyzz_inf(); break;
    default: b = st_rescale(q, z2); break;
  }
  u32 w[32];
  xyzz_store(a, w);
  for (int k = 0; k < 32; k++) a_out[i * 32 + k] = w[k];
  xyzz_store(b, w);
  for (int k = 0; k < 32; k++) b_out[i * 32 + k] = w[k];
}
__global__ __launch_bounds__(256) void k_rowtest_run(const u32* __restrict__ a, const u32* __restrict__ b, size_t n, int dbl_reps, u32* __restrict__ sum_out,
                                                      u32* __restrict__ dbl_out) {
  const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const Lane ln = rowop::lane_init();
  const Pt x = rowop::load(a + i * 32, ln), y = rowop::load(b + i * 32, ln);
  rowop::store(sum_out + i * 32, rowop::add(x, y, ln), ln);
  Pt d = x;
  for (int r = 0; r < dbl_reps; r++) d = rowop::dbl(d, ln);
  rowop::store(dbl_out + i * 32, d, ln);
}
__device__ __forceinline__ bool st_same_point(const Xyzz& u, const Xyzz& v) {
  typedef FqParams P;
  const bool ui = xyzz_is_inf(u), vi = xyzz_is_inf(v);
  if (ui || vi) return ui == vi;
  return fe_eq_canon<P>(fe_reduce<P>(fe_mul<P>(u.X, v.ZZ)), fe_reduce<P>(fe_mul<P>(v.X, u.ZZ))) &&
         fe_eq_canon<P>(fe_reduce<P>(fe_mul<P>(u.Y, v.ZZZ)), fe_reduce<P>(fe_mul<P>(v.Y, u.ZZZ)));
}
__device__ __forceinline__ bool st_record_ok(const u32* w) {            // every coordinate below 2.5 p (top word of 2.5 p: 0x78fac41e)
  for (int c = 0; c < 4; c++) if (w[8 * c + 7] >= 0x78fac41eu) return false;
  return true;
}
__global__ __launch_bounds__(128) void k_rowtest_check(const u32* __restrict__ a, const u32* __restrict__ b, size_t n, int dbl_reps, const u32* __restrict__ sum_out,
                                                        const u32* __restrict__ dbl_out, unsigned long long* __restrict__ mismatches) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[32];
  for (int k = 0; k < 32; k++) w[k] = a[i * 32 + k];
  const Xyzz x = xyzz_load(w);
  for (int k = 0; k < 32; k++) w[k] = b[i * 32 + k];
  const Xyzz y = xyzz_load(w);
  Xyzz d = x;
  for (int r = 0; r < dbl_reps; r++) d = xyzz_dbl(d);
  for (int k = 0; k < 32; k++) w[k] = sum_out[i * 32 + k];
  bool ok = st_record_ok(w) && st_same_point(xyzz_load(w), xyzz_add(x, y));
  for (int k = 0; k < 32; k++) w[k] = dbl_out[i * 32 + k];
  ok = ok && st_record_ok(w) && st_same_point(xyzz_load(w), d);
  if (!ok) atomicAdd(mismatches, 1ull);
}
// wave inversion against the single-lane safegcd and against a * a^-1 == 1: one value per wave
__global__ __launch_bounds__(256) void k_invtest(u64 seed, size_t n, unsigned long long* __restrict__ mismatches) {
  typedef FqParams P;
  const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  u64 s = seed ^ (0x9e3779b97f4a7c15ULL * (i + 1));
  Fq a = st_rand_fq(s);
  const u32 specials[6] = {0u, 1u, 2u, 3u, 0x1fffffffu, 0x10000000u};
  if (i < 6) { a = fe_zero<P>(); a.l[0] = specials[i]; }
  if (i == 6) { a = fe_reduce<P>(fe_neg_canon<P>(fe_one<P>())); }             // -R mod p
  if (i == 7) { a = fe_one<P>(); }
  const Fq r1 = invw::inv<P>(a), r2 = fe_inv_safegcd<P>(a);
  bool ok = fe_eq_canon<P>(fe_reduce<P>(r1), fe_reduce<P>(r2));
  if (!fe_is_zero_canon<P>(fe_reduce<P>(a))) ok = ok && fe_eq_canon<P>(fe_reduce<P>(fe_mul<P>(a, r1)), fe_reduce<P>(fe_one<P>()));
  if (!ok && (threadIdx.x & 63) == 0) atomicAdd(mismatches, 1ull);
}
int selftest_inv_wave_impl(uint64_t seed, size_t n, uint64_t* mismatches_host, hipStream_t s) {
  unsigned long long* cnt;
  MZK_TRY(ws_get(WS_MISC_A, 64, (void**)&cnt));
  MZK_HIP(hipMemsetAsync(cnt, 0, 8, s));
  if (n) hipLaunchKernelGGL(k_invtest, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (u64)seed, n, cnt);
  MZK_HIP(hipGetLastError());
  unsigned long long h = 0;
  MZK_HIP(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, s));
  MZK_HIP(hipStreamSynchronize(s));
  *mismatches_host = (uint64_t)h;
  return MZK_OK;
}
int synth_g1_impl(uint64_t seed, size_t n, void* d_out, hipStream_t s);
int selftest_row_ec_impl(uint64_t seed, size_t n, int dbl_reps, uint64_t* mismatches_host, hipStream_t s) {
  if (n == 0) { *mismatches_host = 0; return MZK_OK; }
  void *plain, *mont, *a, *b, *so, *dd, *cnt;
  MZK_TRY(ws_get(WS_MISC_A, n * 64, &plain));
  MZK_TRY(ws_get(WS_MISC_B, n * 64, &mont));
  MZK_TRY(ws_get(WS_MISC_C, n * 128, &a));
  MZK_TRY(ws_get(WS_MISC_D, n * 128, &b));
  MZK_TRY(ws_get(WS_MISC_E, n * 128, &so));
  MZK_TRY(ws_get(WS_MISC_F, n * 128 + 64, &dd));
  cnt = (char*)dd + n * 128;
  MZK_TRY(synth_g1_impl(seed, n, plain, s));
  MZK_TRY(msm_prepare_points(plain, n, mont, nullptr, s));
  MZK_HIP(hipMemsetAsync(cnt, 0, 8, s));
  hipLaunchKernelGGL(k_rowtest_prepare, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, (const u32*)mont, n, (u64)seed, (u32*)a, (u32*)b);
  hipLaunchKernelGGL(k_rowtest_run, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (const u32*)a, (const u32*)b, n, dbl_reps, (u32*)so, (u32*)dd);
  hipLaunchKernelGGL(k_rowtest_check, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, (const u32*)a, (const u32*)b, n, dbl_reps, (const u32*)so, (const u32*)dd,
                     (unsigned long long*)cnt);
  MZK_HIP(hipGetLastError());
  MZK_HIP(hipMemcpyAsync(mismatches_host, cnt, 8, hipMemcpyDeviceToHost, s));
  MZK_HIP(hipStreamSynchronize(s));
  return MZK_OK;
}

}  // namespace mzk
