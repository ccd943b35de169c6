// mzk_g2.hip -- G2 entry points: the k+1-term G2 MSM of batch_verify_kzg (kzg.rs:114,
// Polynomial::eval_with_powers_on_curve over pk.powers_2) and powers_2 of setup_kzg_with_full_g2 (kzg.rs:42-55).
// Sizes in the reference are tiny, so the shape is the simple one: one lane per (scalar, point) pair does a
// double-and-add on an XYZZ accumulator over Fq2 (mzk_g2.h), one workgroup sums the partial results.
#include "mzk_common.h"
#include "mzk_g2.h"

namespace mzk {

struct Words8k { u32 w[8]; };
__device__ __forceinline__ void ldw(const u32* __restrict__ g, u32* w, int n) { for (int i = 0; i < n; i++) w[i] = g[i]; }

// out[i] = scalars[i] * points[i] as a 64-word XYZZ record; scalars canonicalised like polynomial.rs:162 (sanitize)
__global__ __launch_bounds__(64) void k_g2_pair_mul(const u32* __restrict__ scalars, const u32* __restrict__ points, size_t n, u32* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[8], pw[32], rec[64];
  ldw(scalars + 8 * i, k, 8);
  {
    Fe<FrParams> x = fe_reduce<FrParams>(fe_unpack<FrParams>(k));
    fe_pack<FrParams>(x, k);
  }
  ldw(points + 32 * i, pw, 32);
  Xyzz2 r = x2_inf();
  if (!g2_words_is_inf(pw)) r = x2_scalar_mul(g2_load_plain(pw), k);
  x2_store(r, rec);
  for (int j = 0; j < 64; j++) out[64 * i + j] = rec[j];
}
// sum of n XYZZ records -> one affine wire point (32 words)
constexpr int G2_SUM_THREADS = 128;
__global__ __launch_bounds__(G2_SUM_THREADS) void k_g2_sum(const u32* __restrict__ recs, size_t n, u32* __restrict__ out) {
  __shared__ u32 sh[G2_SUM_THREADS * 64];
  Xyzz2 acc = x2_inf();
  for (size_t i = threadIdx.x; i < n; i += G2_SUM_THREADS) {
    u32 rec[64];
    ldw(recs + 64 * i, rec, 64);
    acc = x2_add(acc, x2_load(rec));
  }
  {
    u32 rec[64];
    x2_store(acc, rec);
    for (int j = 0; j < 64; j++) sh[64 * threadIdx.x + j] = rec[j];
  }
  __syncthreads();
  for (int off = G2_SUM_THREADS / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      u32 a[64], b[64];
      for (int j = 0; j < 64; j++) { a[j] = sh[64 * threadIdx.x + j]; b[j] = sh[64 * (threadIdx.x + off) + j]; }
      u32 rec[64];
      x2_store(x2_add(x2_load(a), x2_load(b)), rec);
      for (int j = 0; j < 64; j++) sh[64 * threadIdx.x + j] = rec[j];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    u32 rec[64], w[32];
    for (int j = 0; j < 64; j++) rec[j] = sh[j];
    g2_store_plain(x2_load(rec), w);
    for (int j = 0; j < 32; j++) out[j] = w[j];
  }
}
// powers[i] = alpha^(first + i) * g2, affine wire points
__global__ __launch_bounds__(64) void k_g2_powers(Words8k alpha_plain, const u32* __restrict__ g2_words, size_t first, size_t count, u32* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  typedef FrParams R;
  const Fe<R> a = fe_to_mont<R>(fe_unpack<R>(alpha_plain.w));
  u32 k[8], gw[32], w[32];
  fe_pack<R>(fe_from_mont<R>(fe_pow_u64<R>(a, (u64)(first + i))), k);
  ldw(g2_words, gw, 32);
  Xyzz2 r = x2_inf();
  if (!g2_words_is_inf(gw)) r = x2_scalar_mul(g2_load_plain(gw), k);
  g2_store_plain(r, w);
  for (int j = 0; j < 32; j++) out[32 * i + j] = w[j];
}

int g2_msm_dev_impl(const void* d_scalars, const void* d_points, size_t n, void* d_out, hipStream_t s) {
  if (!d_out || ((!d_scalars || !d_points) && n)) { set_error("msm_g2: null pointer"); return MZK_E_ARG; }
  if (n == 0) { MZK_HIP(hipMemsetAsync(d_out, 0, 128, s)); return MZK_OK; }     // empty polynomial -> infinity (polynomial.rs:160)
  u32* recs;
  MZK_TRY(ws_get(WS_XYZZ_TMP, n * 256, (void**)&recs));
  hipLaunchKernelGGL(k_g2_pair_mul, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (const u32*)d_scalars, (const u32*)d_points, n, recs);
  hipLaunchKernelGGL(k_g2_sum, dim3(1), dim3(G2_SUM_THREADS), 0, s, (const u32*)recs, n, (u32*)d_out);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
}  // namespace mzk

using namespace mzk;
extern "C" {

int mzk_msm_g2_bn254_dev(const void* d_scalars, const void* d_points_xy, size_t n, void* d_out_xy, void* stream) {
  MZK_ENTER();
  WsGuard wsg((hipStream_t)stream);
  return g2_msm_dev_impl(d_scalars, d_points_xy, n, d_out_xy, (hipStream_t)stream);
}
int mzk_msm_g2_bn254(const uint64_t* scalars, const uint64_t* points_xy, size_t n, uint64_t out_xy[16]) {
  MZK_ENTER();
  if (!out_xy || ((!scalars || !points_xy) && n)) { set_error("msm_g2: null pointer"); return MZK_E_ARG; }
  const HostField* fq = host_field(MZK_FIELD_FQ);
  for (size_t i = 0; i < 4 * n; i++)
    if (!h_is_canonical(fq, points_xy + 4 * i)) { set_error("msm_g2: point coordinate %zu not canonical", i); return MZK_E_RANGE; }
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  void *d_s, *d_p, *d_o;
  MZK_TRY(ws_get(WS_MSM_SCALARS, n ? n * 32 : 16, &d_s));
  MZK_TRY(ws_get(WS_MSM_POINTS, n ? n * 128 : 16, &d_p));
  MZK_TRY(ws_get(WS_MSM_OUT, 4096, &d_o));
  if (n) {
    MZK_HIP(hipMemcpyAsync(d_s, scalars, n * 32, hipMemcpyHostToDevice, s));
    MZK_HIP(hipMemcpyAsync(d_p, points_xy, n * 128, hipMemcpyHostToDevice, s));
  }
  MZK_TRY(g2_msm_dev_impl(d_s, d_p, n, d_o, s));
  MZK_TRY(d2h_sync(out_xy, d_o, 128, s));
  return MZK_OK;
}
int mzk_kzg_setup_g2(const uint64_t alpha[4], const uint64_t g2_xy[16], size_t max_d, uint64_t* powers2_xy) {
  MZK_ENTER();
  if (!alpha || !g2_xy || !powers2_xy) { set_error("kzg_setup_g2: null pointer"); return MZK_E_ARG; }
  const HostField* fq = host_field(MZK_FIELD_FQ);
  if (!h_is_canonical(host_field(MZK_FIELD_FR), alpha)) { set_error("kzg_setup_g2: alpha not canonical"); return MZK_E_RANGE; }
  for (int i = 0; i < 4; i++) if (!h_is_canonical(fq, g2_xy + 4 * i)) { set_error("kzg_setup_g2: generator not canonical"); return MZK_E_RANGE; }
  const size_t count = max_d + 1;                       // `for _ in 0..1 + max_d` (kzg.rs:48)
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  void *d_g, *d_o;
  MZK_TRY(ws_get(WS_MSM_OUT, 4096, &d_g));
  MZK_TRY(ws_get(WS_MSM_POINTS, count * 128, &d_o));
  MZK_HIP(hipMemcpyAsync(d_g, g2_xy, 128, hipMemcpyHostToDevice, s));
  Words8k aw;
  for (int i = 0; i < 4; i++) { aw.w[2 * i] = (u32)alpha[i]; aw.w[2 * i + 1] = (u32)(alpha[i] >> 32); }
  hipLaunchKernelGGL(k_g2_powers, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s, aw, (const u32*)d_g, (size_t)0, count, (u32*)d_o);
  MZK_HIP(hipGetLastError());
  MZK_TRY(d2h_sync(powers2_xy, d_o, count * 128, s));
  return MZK_OK;
}

}  // extern "C"
