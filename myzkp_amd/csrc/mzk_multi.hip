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
  MZK_TRY(d2h_sync(out_xy, d_out, 64, s));
  return MZK_OK;
}


// ---- one transform sharded over the contexts (the four-step layout; schedule and layouts as in
// myzkp_amd/sharded.py, which runs the same steps one process per GPU over RCCL's all-to-all) ---------------------------
// Context r holds part r of the vector.  An "exchange" is W x W chunk copies (peer copies between GPUs, device copies when
// contexts share one): destination d pulls chunk d of every source's buffer on ITS OWN stream after waiting for the event
// the source recorded behind the step that produced the buffer.  Three scratch buffers per context (A, B, C: n / W elements
// each) are each written once per phase and only re-used after every reader is known to be done (the waits above).
struct Exchange {
  int W;
  size_t chunk_bytes;
  hipEvent_t ev[MZK_MAX_CTX];
};
static int record_all(Exchange& x) {
  for (int r = 0; r < x.W; r++) {
    CtxScope sc(r);
    if (!sc.ok) return MZK_E_ARG;
    MZK_HIP(hipEventRecord(x.ev[r], ctx().stream));
  }
  return MZK_OK;
}
// dst[d] + s * chunk  <-  src[s] + d * chunk, for every (d, s); wait_events: the sources were produced by this call.
// Destination d drives ALL its inbound links at once: the pull from source (d + k) % W runs on d's k-th exchange stream
// (its own chunk, k = 0, on the context's stream), so the W - 1 peer copies of a GPU overlap each other, and the
// staggered start means that at any moment every source is being read by a different destination (xGMI is point to
// point: W - 1 links per GPU, each carrying one chunk).  The exchange streams start behind the context's stream (xready:
// the destination buffer's previous readers are done) and the context's stream continues behind all of them (xdone).
static int exchange(Exchange& x, void* const* src, void* const* dst, bool wait_events) {
  for (int d = 0; d < x.W; d++) {
    CtxScope sc(d);
    if (!sc.ok) return MZK_E_ARG;
    Context& c = ctx();
    hipStream_t st = c.stream;
    const int dev_d = c.device;
    if (!c.xready) MZK_HIP(hipEventCreateWithFlags(&c.xready, hipEventDisableTiming));
    MZK_HIP(hipEventRecord(c.xready, st));
    for (int k = 0; k < x.W; k++) {
      const int s = (d + k) % x.W;
      hipStream_t xs = st;
      if (k) {
        if (!c.xstream[k]) MZK_HIP(hipStreamCreateWithFlags(&c.xstream[k], hipStreamNonBlocking));
        if (!c.xdone[k]) MZK_HIP(hipEventCreateWithFlags(&c.xdone[k], hipEventDisableTiming));
        xs = c.xstream[k];
        MZK_HIP(hipStreamWaitEvent(xs, c.xready, 0));
        if (wait_events) MZK_HIP(hipStreamWaitEvent(xs, x.ev[s], 0));
      }
      const char* from = (const char*)src[s] + (size_t)d * x.chunk_bytes;
      char* to = (char*)dst[d] + (size_t)s * x.chunk_bytes;
      const int dev_s = mzk_ctx_device(s);
      if (dev_s == dev_d) MZK_HIP(hipMemcpyAsync(to, from, x.chunk_bytes, hipMemcpyDeviceToDevice, xs));
      else MZK_HIP(hipMemcpyPeerAsync(to, dev_d, from, dev_s, x.chunk_bytes, xs));
      if (k) MZK_HIP(hipEventRecord(c.xdone[k], xs));
    }
    for (int k = 1; k < x.W; k++) MZK_HIP(hipStreamWaitEvent(st, c.xdone[k], 0));
  }
  return MZK_OK;
}
static int ntt_multi_impl(int fid, const uint64_t* root, const void* const* in, void* const* out, size_t n, int inverse, int layout_in, int layout_out) {
  if (fid != MZK_FIELD_FR && fid != MZK_FIELD_M128) { set_error("ntt_multi: field id %d has no NTT on this path", fid); return MZK_E_ARG; }
  if (n == 0) return MZK_OK;
  if (n & (n - 1)) { set_error("cannot compute ntt of non-power-of-two sequence"); return MZK_E_NOT_POW2; }
  if (!root || !in || !out) { set_error("ntt_multi: null pointer"); return MZK_E_ARG; }
  const int W = ctx_count();
  if ((W & (W - 1)) || (size_t)W * W > n || W > 16) { set_error("ntt_multi: %d contexts: need a power of two <= 16 with world^2 <= n", W); return MZK_E_ARG; }
  const bool cyc_in = layout_in == MZK_LAYOUT_CYCLIC, cyc_out = layout_out == MZK_LAYOUT_CYCLIC;
  if ((layout_in != MZK_LAYOUT_CONTIGUOUS && !cyc_in) || (layout_out != MZK_LAYOUT_CONTIGUOUS && !cyc_out) || (cyc_in && cyc_out)) {
    set_error("ntt_multi: layouts are contiguous->contiguous, contiguous->cyclic or cyclic->contiguous"); return MZK_E_ARG;
  }
  for (int r = 0; r < W; r++) if (!in[r] || !out[r]) { set_error("ntt_multi: null part pointer for context %d", r); return MZK_E_ARG; }
  const HostField* hf = host_field(fid);
  if (!h_is_canonical(hf, root)) { set_error("ntt: root not canonical"); return MZK_E_RANGE; }
  const size_t esz = field_bytes(fid), m = n / (size_t)W, cols = m / (size_t)W;
  if (W == 1) {
    CtxScope sc(0);
    if (!sc.ok) return MZK_E_ARG;
    WsGuard wsg(ctx().stream);
    MZK_TRY(ntt_dev_impl(fid, root, in[0], out[0], n, inverse, nullptr, ctx().stream));
    MZK_HIP(hipStreamSynchronize(ctx().stream));
    return MZK_OK;
  }
  // the reference's two assertions on the root (ntt.rs:15-22), before anything is enqueued
  uint64_t t[4], w[4] = {0, 0, 0, 0}, root_W[4], root_m[4];
  h_powmod_u64(hf, t, root, n);
  if (!h_is_one(hf, t)) { set_error("primitive root must be nth root of unity, where n is len(values)"); return MZK_E_ROOT_ORDER; }
  h_powmod_u64(hf, t, root, n / 2);
  if (h_is_one(hf, t)) { set_error("primitive root is not primitive nth root of unity, where n is len(values)"); return MZK_E_ROOT_PRIM; }
  h_powmod_u64(hf, root_W, root, m);            // forward roots of the W- and m-point transforms, as their callers pass them
  h_powmod_u64(hf, root_m, root, (uint64_t)W);
  if (inverse) h_powmod_u64(hf, w, root, n - 1); else memcpy(w, root, 8 * hf->nl);      // the root the sums run over

  Exchange x{W, cols * esz, {}};
  void *A[MZK_MAX_CTX], *B[MZK_MAX_CTX], *C[MZK_MAX_CTX];
  int rc = MZK_OK;
  int made = 0;
  for (int r = 0; r < W && rc == MZK_OK; r++) {
    CtxScope sc(r);
    if (!sc.ok) { rc = MZK_E_ARG; break; }
    if (hipEventCreateWithFlags(&x.ev[r], hipEventDisableTiming) != hipSuccess) { set_error("ntt_multi: hipEventCreate failed"); rc = MZK_E_HIP; break; }
    made = r + 1;
    WsGuard wsg(ctx().stream);      // the exchanges below write these slots on the context's stream: order them behind whoever used them last
    rc = ws_get(WS_MISC_D, m * esz, &A[r]);
    if (rc == MZK_OK) rc = ws_get(WS_MISC_E, m * esz, &B[r]);
    if (rc == MZK_OK) rc = ws_get(WS_MISC_F, m * esz, &C[r]);
  }
  // a local step on every context: fn(r, stream)
  auto each = [&](auto fn) -> int {
    for (int r = 0; r < W; r++) {
      CtxScope sc(r);
      if (!sc.ok) return MZK_E_ARG;
      WsGuard wsg(ctx().stream);
      MZK_TRY(fn(r, ctx().stream));
    }
    return MZK_OK;
  };
  auto twiddle = [&](int r, uint64_t* o) { h_powmod_u64(hf, o, w, (uint64_t)r); };
  if (rc == MZK_OK && !cyc_in) {
    // contiguous in: exchange, W-point transforms across the ranks, exchange, local transform with its twiddle fused
    // (fast_coset_evaluate with offset w^rank, ntt.rs:254-269), [exchange + interleave for a contiguous result]
    rc = exchange(x, (void* const*)in, A, false);
    if (rc == MZK_OK) rc = each([&](int r, hipStream_t s) { return ntt_columns_dev_impl(fid, root_W, A[r], B[r], (size_t)W, cols, inverse, s); });
    if (rc == MZK_OK) rc = record_all(x);
    if (rc == MZK_OK) rc = exchange(x, B, C, true);
    if (rc == MZK_OK) rc = each([&](int r, hipStream_t s) -> int {
      uint64_t tw[4] = {0, 0, 0, 0};
      twiddle(r, tw);
      void* dst = cyc_out ? out[r] : A[r];
      if (!inverse) return coset_lde_dev_impl(fid, C[r], m, tw, root_m, dst, m, s);
      MZK_TRY(poly_scale_dev_impl(fid, C[r], m, tw, nullptr, C[r], s));
      return ntt_dev_impl(fid, root_m, C[r], dst, m, 1, nullptr, s);
    });
    if (rc == MZK_OK && !cyc_out) {
      rc = record_all(x);
      if (rc == MZK_OK) rc = exchange(x, A, B, true);
      if (rc == MZK_OK) rc = each([&](int r, hipStream_t s) { return transpose_elems_dev_impl(fid, B[r], out[r], (size_t)W, cols, s); });
    }
  } else if (rc == MZK_OK) {
    // cyclic in: local transform, Polynomial::scale by w^rank (polynomial.rs:167-174), exchange, W-point transforms, exchange
    rc = each([&](int r, hipStream_t s) -> int {
      uint64_t tw[4] = {0, 0, 0, 0};
      twiddle(r, tw);
      MZK_TRY(ntt_dev_impl(fid, root_m, in[r], A[r], m, inverse, nullptr, s));
      return poly_scale_dev_impl(fid, A[r], m, tw, nullptr, A[r], s);
    });
    if (rc == MZK_OK) rc = record_all(x);
    if (rc == MZK_OK) rc = exchange(x, A, B, true);
    if (rc == MZK_OK) rc = each([&](int r, hipStream_t s) { return ntt_columns_dev_impl(fid, root_W, B[r], C[r], (size_t)W, cols, inverse, s); });
    if (rc == MZK_OK) rc = record_all(x);
    if (rc == MZK_OK) rc = exchange(x, C, (void* const*)out, true);
  }
  for (int r = 0; r < made; r++) {
    CtxScope sc(r);
    if (sc.ok) { (void)hipStreamSynchronize(ctx().stream); (void)hipEventDestroy(x.ev[r]); }
  }
  return rc;
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
  MZK_ENTER();
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
  MZK_ENTER();
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
  MZK_ENTER();
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
  MZK_ENTER();
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

// One n-point transform (ntt / intt, ntt.rs:7-64) whose vector is spread over the contexts; see include/mzk.h.
int mzk_ntt_multi_dev(int field_id, const uint64_t* root, const void* const* d_in_parts, void* const* d_out_parts, size_t n, int inverse,
                      int layout_in, int layout_out) {
  MZK_ENTER();
  return ntt_multi_impl(field_id, root, d_in_parts, d_out_parts, n, inverse, layout_in, layout_out);
}
// host vector in natural order: context r gets x[r n/W, (r+1) n/W) and returns the same slice of the result
int mzk_ntt_multi(int field_id, const uint64_t* root, const uint64_t* in, uint64_t* out, size_t n, int inverse) {
  MZK_ENTER();
  if (n == 0) return MZK_OK;
  if (!in || !out) { set_error("ntt_multi: null pointer"); return MZK_E_ARG; }
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("ntt_multi: field id %d has no NTT on this path", field_id); return MZK_E_ARG; }
  const int W = ctx_count();
  if (n % (size_t)W) { set_error("ntt_multi: %d contexts do not divide n", W); return MZK_E_ARG; }
  const size_t esz = field_bytes(field_id), m = n / (size_t)W;
  void *din[MZK_MAX_CTX], *dout[MZK_MAX_CTX];
  for (int r = 0; r < W; r++) {
    CtxScope sc(r);
    if (!sc.ok) return MZK_E_ARG;
    hipStream_t s = ctx().stream;
    WsGuard wsg(s);
    MZK_TRY(ws_get(WS_NTT_IO_B, m * esz, &din[r]));
    MZK_TRY(ws_get(WS_MISC_C, m * esz, &dout[r]));
    MZK_HIP(hipMemcpyAsync(din[r], (const char*)in + (size_t)r * m * esz, m * esz, hipMemcpyHostToDevice, s));
  }
  for (int r = 0; r < W; r++) { CtxScope sc(r); if (!sc.ok) return MZK_E_ARG; MZK_HIP(hipStreamSynchronize(ctx().stream)); }
  MZK_TRY(ntt_multi_impl(field_id, root, (const void* const*)din, dout, n, inverse, MZK_LAYOUT_CONTIGUOUS, MZK_LAYOUT_CONTIGUOUS));
  for (int r = 0; r < W; r++) {
    CtxScope sc(r);
    if (!sc.ok) return MZK_E_ARG;
    MZK_HIP(hipMemcpyAsync((char*)out + (size_t)r * m * esz, dout[r], m * esz, hipMemcpyDeviceToHost, ctx().stream));
    MZK_HIP(hipStreamSynchronize(ctx().stream));
  }
  return MZK_OK;
}

}  // extern "C"
