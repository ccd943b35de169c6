// mzk_multi.hip -- the MSM / KZG commit sharded over several GPUs from ONE process, inside the C ABI.
//
// SURVEY 8e / BASELINE configs[3]: sum_i s_i P_i is a sum over independent pairs.  Context r of W (mzk_init_devices)
// owns the contiguous slice [lo_r, hi_r) of the scalar and point arrays, reduces it to ONE 128-byte XYZZ partial
// with the full single-GPU pipeline on its own stream and workspace, and the only exchange is the gather of the W
// partial records onto context 0, which folds them (k_fold_partials) and converts to affine once.  The records
// travel through a pinned host buffer: 128 bytes per GPU need neither peer access nor a collective library, and all
// W pipelines run concurrently because every enqueue below is asynchronous.  (One-process-per-GPU jobs use the
// *_partial_dev entry points with an RCCL all-gather instead: myzkp_amd/sharded.py, bench.py.)
//
// Several contexts may name the same device ordinal, so the whole path -- W streams, W workspaces, gather, fold --
// runs on a one-GPU box too (tests/cpp/test_multi_device.cpp, tests/test_gpu_multi.py).
#include "mzk_common.h"

using namespace mzk;

struct mzk_srs_multi {
  int world;
  size_t n;
  size_t lo[MZK_MAX_CTX + 1];
  mzk_srs* shard[MZK_MAX_CTX];
};

namespace mzk {

static void shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  const size_t base = n / (size_t)world, extra = n % (size_t)world;
  const size_t r = (size_t)rank;
  *lo = r * base + (r < extra ? r : extra);
  *hi = *lo + base + (r < extra ? 1 : 0);
}

// pinned landing zone of the partial records (one per context), created on first use
static uint64_t* g_pinned = nullptr;
static int pinned_records(uint64_t** out) {
  if (!g_pinned) MZK_HIP(hipHostMalloc((void**)&g_pinned, (size_t)MZK_MAX_CTX * 128, hipHostMallocPortable));
  *out = g_pinned;
  return MZK_OK;
}

// wait for every context's stream, then fold the W records on context 0 and return the affine point
static int gather_and_fold(int world, const uint64_t* h_records, uint64_t out_xy[8]) {
  for (int r = 0; r < world; r++) {
    CtxScope sc(r);
    if (!sc.ok) return MZK_E_ARG;
    MZK_HIP(hipStreamSynchronize(ctx().stream));
  }
  CtxScope sc(0);
  if (!sc.ok) return MZK_E_ARG;
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  void* d_rec;
  MZK_TRY(ws_get(WS_MISC_E, (size_t)world * 128 + 64, &d_rec));
  MZK_HIP(hipMemcpyAsync(d_rec, h_records, (size_t)world * 128, hipMemcpyHostToDevice, s));
  void* d_out = (char*)d_rec + (size_t)world * 128;
  MZK_TRY(msm_fold_partials_impl(d_rec, world, d_out, s));
  MZK_HIP(hipMemcpyAsync(out_xy, d_out, 64, hipMemcpyDeviceToHost, s));
  MZK_HIP(hipStreamSynchronize(s));
  return MZK_OK;
}

}  // namespace mzk

extern "C" {

void mzk_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  if (world < 1 || rank < 0 || rank >= world) { if (lo) *lo = 0; if (hi) *hi = 0; return; }
  size_t a, b;
  shard_range(n, rank, world, &a, &b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

int mzk_msm_g1_bn254_multi(const uint64_t* scalars, const uint64_t* points_xy, size_t n, uint64_t out_xy[8]) {
  MZK_TRY(ensure_init());
  if (!out_xy || ((!scalars || !points_xy) && n)) { set_error("msm_multi: null pointer"); return MZK_E_ARG; }
  const int world = ctx_count();
  uint64_t* h_rec;
  MZK_TRY(pinned_records(&h_rec));
  for (int r = 0; r < world; r++) {
    size_t lo, hi;
    shard_range(n, r, world, &lo, &hi);
    CtxScope sc(r);
    if (!sc.ok) return MZK_E_ARG;
    hipStream_t s = ctx().stream;
    WsGuard wsg(s);
    void *d_s, *d_p, *d_o;
    const size_t m = hi - lo;
    MZK_TRY(ws_get(WS_MSM_SCALARS, m ? m * 32 : 16, &d_s));
    MZK_TRY(ws_get(WS_MISC_A, m ? m * 64 : 16, &d_p));
    MZK_TRY(ws_get(WS_MISC_B, 256, &d_o));
    if (m) {
      MZK_HIP(hipMemcpyAsync(d_s, scalars + 4 * lo, m * 32, hipMemcpyHostToDevice, s));
      MZK_HIP(hipMemcpyAsync(d_p, points_xy + 8 * lo, m * 64, hipMemcpyHostToDevice, s));
    }
    MZK_TRY(msm_dev_impl(d_s, d_p, m, MSM_PTS_PLAIN, 0, d_o, true, s));
    MZK_HIP(hipMemcpyAsync(h_rec + 16 * r, d_o, 128, hipMemcpyDeviceToHost, s));
  }
  return gather_and_fold(world, h_rec, out_xy);
}

void mzk_srs_multi_free(mzk_srs_multi* h) {
  if (!h) return;
  for (int r = 0; r < h->world; r++) mzk_srs_free(h->shard[r]);
  delete h;
}

static mzk_srs_multi* new_multi(size_t n) {
  mzk_srs_multi* h = new mzk_srs_multi();
  h->world = ctx_count();
  h->n = n;
  for (int r = 0; r < MZK_MAX_CTX; r++) h->shard[r] = nullptr;
  for (int r = 0; r < h->world; r++) {
    size_t hi;
    shard_range(n, r, h->world, &h->lo[r], &hi);
    h->lo[r + 1] = hi;
  }
  return h;
}

int mzk_srs_upload_multi(const uint64_t* powers_xy, size_t n, mzk_srs_multi** out) {
  MZK_TRY(ensure_init());
  if (!out || (!powers_xy && n)) { set_error("srs_upload_multi: null pointer"); return MZK_E_ARG; }
  mzk_srs_multi* h = new_multi(n);
  for (int r = 0; r < h->world; r++) {
    CtxScope sc(r);
    int rc = sc.ok ? mzk_srs_upload(powers_xy + 8 * h->lo[r], h->lo[r + 1] - h->lo[r], &h->shard[r]) : MZK_E_ARG;
    if (rc != MZK_OK) { mzk_srs_multi_free(h); return rc; }
  }
  *out = h;
  return MZK_OK;
}

// setup_kzg (kzg.rs:27-40) sharded: context r builds powers [lo_r, hi_r) of [alpha^i] g1 on its own GPU and keeps
// them as its SRS shard -- no point ever crosses PCIe or xGMI.
int mzk_kzg_setup_srs_multi(const uint64_t alpha[4], const uint64_t g1_xy[8], size_t max_d, int with_tables, mzk_srs_multi** out) {
  MZK_TRY(ensure_init());
  if (!out || !alpha || !g1_xy) { set_error("setup_srs_multi: null pointer"); return MZK_E_ARG; }
  mzk_srs_multi* h = new_multi(max_d + 1);
  void* tmp[MZK_MAX_CTX] = {};
  int rc = MZK_OK;
  // enqueue every context's setup first (they run concurrently), then build the handles (each synchronises its stream)
  for (int r = 0; r < h->world && rc == MZK_OK; r++) {
    CtxScope sc(r);
    if (!sc.ok) { rc = MZK_E_ARG; break; }
    const size_t cnt = h->lo[r + 1] - h->lo[r];
    if (hipMalloc(&tmp[r], cnt ? cnt * 64 : 64) != hipSuccess) { set_error("setup_srs_multi: hipMalloc failed"); rc = MZK_E_HIP; break; }
    WsGuard wsg(ctx().stream);
    rc = kzg_setup_g1_dev(alpha, g1_xy, h->lo[r], cnt, tmp[r], ctx().stream);
  }
  for (int r = 0; r < h->world && rc == MZK_OK; r++) {
    CtxScope sc(r);
    if (!sc.ok) { rc = MZK_E_ARG; break; }
    rc = mzk_srs_from_device_ex(tmp[r], h->lo[r + 1] - h->lo[r], with_tables, &h->shard[r], ctx().stream);
  }
  for (int r = 0; r < h->world; r++) if (tmp[r]) { CtxScope sc(r); (void)hipStreamSynchronize(ctx().stream); (void)hipFree(tmp[r]); }
  if (rc != MZK_OK) { mzk_srs_multi_free(h); return rc; }
  *out = h;
  return MZK_OK;
}

size_t mzk_srs_multi_shard_lo(const mzk_srs_multi* h, int rank) { return (h && rank >= 0 && rank <= h->world) ? h->lo[rank] : 0; }
int mzk_srs_multi_world(const mzk_srs_multi* h) { return h ? h->world : 0; }

// commit_kzg (kzg.rs:57-59) over the sharded SRS.  host_coef != NULL: coefficients in host memory (each context
// copies its slice); otherwise d_coef_shards[r] is a device pointer ON CONTEXT r's GPU to coefficients
// [lo_r, min(hi_r, n)) -- already complete when the call is made (the call does not know the producer's stream).
static int commit_multi(const mzk_srs_multi* h, const uint64_t* host_coef, const void* const* d_coef_shards, size_t n, uint64_t out_xy[8]) {
  MZK_TRY(ensure_init());
  if (!h || !out_xy || (n && !host_coef && !d_coef_shards)) { set_error("commit_srs_multi: null pointer"); return MZK_E_ARG; }
  if (h->world != ctx_count()) { set_error("commit_srs_multi: handle was built for %d contexts, %d are initialised", h->world, ctx_count()); return MZK_E_ARG; }
  if (n > h->n) { set_error("index out of bounds: the len is %zu but the index is %zu", h->n, h->n); return MZK_E_LENGTH; }   // powers[i], polynomial.rs:162
  uint64_t* h_rec;
  MZK_TRY(pinned_records(&h_rec));
  for (int r = 0; r < h->world; r++) {
    const size_t lo = h->lo[r], hi = h->lo[r + 1] < n ? h->lo[r + 1] : n;
    const size_t m = hi > lo ? hi - lo : 0;
    CtxScope sc(r);
    if (!sc.ok) return MZK_E_ARG;
    hipStream_t s = ctx().stream;
    WsGuard wsg(s);
    void *d_s = nullptr, *d_o;
    MZK_TRY(ws_get(WS_MISC_B, 256, &d_o));
    if (host_coef) {
      MZK_TRY(ws_get(WS_MSM_SCALARS, m ? m * 32 : 16, &d_s));
      if (m) MZK_HIP(hipMemcpyAsync(d_s, host_coef + 4 * lo, m * 32, hipMemcpyHostToDevice, s));
    } else if (m) {
      d_s = (void*)d_coef_shards[r];
      if (!d_s) { set_error("commit_srs_multi_dev: null shard pointer for context %d", r); return MZK_E_ARG; }
    }
    const mzk_srs* sh = h->shard[r];
    MZK_TRY(msm_dev_impl(d_s ? d_s : d_o, sh->d_points_mont, m, sh->kind(), sh->n, d_o, true, s));
    MZK_HIP(hipMemcpyAsync(h_rec + 16 * r, d_o, 128, hipMemcpyDeviceToHost, s));
  }
  return gather_and_fold(h->world, h_rec, out_xy);
}
int mzk_kzg_commit_srs_multi(const mzk_srs_multi* h, const uint64_t* coef, size_t n, uint64_t out_xy[8]) {
  if (!coef && n) { set_error("commit_srs_multi: null pointer"); return MZK_E_ARG; }
  return commit_multi(h, coef ? coef : (const uint64_t*)out_xy, nullptr, n, out_xy);
}
int mzk_kzg_commit_srs_multi_dev(const mzk_srs_multi* h, const void* const* d_coef_shards, size_t n, uint64_t out_xy[8]) {
  return commit_multi(h, nullptr, d_coef_shards, n, out_xy);
}

}  // extern "C"
