// mzk_common.h -- process context, error reporting, workspace and host-side parameter math shared by
// the library's translation units.  One process drives one GPU (one rank per GPU under torch.distributed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>
#include "../../include/mzk.h"
#include "mzk_field.h"

namespace mzk {

// ---- tuning switches --------------------------------------------------------------------------------
// The SHIPPED library reads no environment variable: a caller's environment must not be able to change the code path.  The
// A/B switches of tools/timing and tools/gpu_jobs exist only in the tuning build (python -m myzkp_amd.build --tuning ->
// myzkp_amd/libmzk_hip_tuning.so, compiled with -DMZK_TUNING, loaded through MZK_HIP_LIB); there tune_int reads MZK_<NAME>, here it
// is the constant default, and the kernels only a non-default switch can reach are not compiled at all.
#ifdef MZK_TUNING
#include <stdlib.h>
static inline int tune_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; } static inline const char* tune_str(const char* name) { return getenv(name); }
#else
static inline constexpr int tune_int(const char*, int dflt) { return dflt; }
static inline constexpr const char* tune_str(const char*) { return nullptr; }
#endif

// ---- error plumbing ------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
void clear_error();                            // a refused allocation that the call then does without leaves no message behind
size_t table_budget_bytes();                   // mzk_set_table_budget (0 = no limit)
int hip_fail(hipError_t e, const char* what, const char* file, int line);
#define MZK_HIP(x)                                                        \
  do {                                                                    \
    hipError_t _e = (x);                                                  \
    if (_e != hipSuccess) return mzk::hip_fail(_e, #x, __FILE__, __LINE__); \
  } while (0)
#define MZK_TRY(x)        \
  do {                    \
    int _rc = (x);        \
    if (_rc != MZK_OK) return _rc; \
  } while (0)

// ---- contexts -----------------------------------------------------------------------------------------
// A context = (device ordinal, library-owned stream, grow-only workspace, cached tables).  mzk_init creates one;
// mzk_init_devices creates one per listed ordinal (duplicates allowed: several contexts may share one GPU, which is
// how the multi-GPU path is exercised on a one-GPU box).  Every entry point works on the CURRENT context
// (mzk_ctx_select; context 0 after init); the *_multi entry points walk all of them.
constexpr int MZK_MAX_CTX = 16;
struct WsBuf { void* p = nullptr; size_t cap = 0; uint64_t epoch = 0; };      // epoch: the outermost API call that last asked for the slot
enum WsSlot { WS_NTT_TMP = 0, WS_NTT_IO_A, WS_NTT_IO_B, WS_MSM_POINTS, WS_MSM_SCALARS, WS_MSM_COUNTS, WS_MSM_OFFSETS,
              WS_MSM_CURSOR, WS_MSM_ENTRIES, WS_MSM_BUCKETS, WS_MSM_RED_A, WS_MSM_RED_B, WS_MSM_SCAN, WS_MSM_OUT, WS_MSM_SLOTS, WS_MSM_WGHIST, WS_BATCHINV, WS_XYZZ_TMP, WS_MERKLE_NODES, WS_FB_TABLE16, WS_FB_TABLE8, WS_NTT_PRE, WS_NTT_PRE_M128,
              WS_MISC_A, WS_MISC_B, WS_MISC_C, WS_MISC_D, WS_MISC_E, WS_MISC_F, WS_COUNT };
enum AttrFlag { ATTR_FINE_SCATTER4 = 0, ATTR_FINE_SCATTER8, ATTR_DIGITS_LDS, ATTR_SMALL_MSM, ATTR_NTT_LARGE_FR, ATTR_NTT_LARGE_M128, ATTR_NTT_LARGE_M128_2WG, ATTR_MANY_SORT1, ATTR_TAIL_ROW, ATTR_COARSE_STAGED4, ATTR_COARSE_STAGED8, ATTR_COUNT };
struct Context {
  bool ready = false;
  int index = 0;                 // position in the context table (keys the per-context caches of the other translation units)
  int device = -1;
  int num_cu = 256;
  hipStream_t stream = nullptr;  // library-owned stream for the host-buffer entry points
  WsBuf ws[WS_COUNT];
  uint64_t ws_gen = 1;           // bumps whenever every workspace buffer has been released (cached device tables die with it)
  // Stream-order guard of the shared workspace: ws_last = stream of the last entry point; an entry point on another
  // stream records ws_event on ws_last and waits for it before touching the slots (WsGuard).
  hipEvent_t ws_event = nullptr;
  hipEvent_t fork_event = nullptr, join_event = nullptr;     // mzk_kzg_commit_srs_batch_dev: caller stream -> lanes -> caller stream
  hipStream_t ws_last = nullptr;
  bool ws_used = false;
  bool attr_done[ATTR_COUNT] = {};   // hipFuncSetAttribute(MaxDynamicSharedMemorySize) applied on this context's device
  // mzk_ntt_multi's all-to-all: destination context pulls its chunk of every source on a stream of its own per source
  // (created on first use), so all inbound xGMI links of the GPU carry data at once; xready orders them behind the
  // context's stream, xdone[k] joins them back.
  hipStream_t xstream[MZK_MAX_CTX] = {};
  hipEvent_t xdone[MZK_MAX_CTX] = {};
  hipEvent_t xready = nullptr;
  void* bounce = nullptr;        // SMALL_D2H_MAX bytes of pinned host memory: the landing zone of d2h_sync's small results
};
// Result of a call -> host memory, and wait: the stream is idle on return.  A commitment, a Merkle root, a handful of lengths are 32-128
// bytes, and hipMemcpyAsync into PAGEABLE memory costs ~12 us more than the same copy into pinned memory before the synchronize returns
// (tools/microbench/root_mailbox.hip: 73.8 against 63.6 us around a 50-us kernel; a host-mapped mailbox the host polls saves 4 more and was
// not worth its moving parts) -- per FRI round, per small commitment.  Up to SMALL_D2H_MAX bytes land in the context's pinned buffer and
// are copied on from there; larger results go straight to the caller's buffer as before (56 GB/s either way).
constexpr size_t SMALL_D2H_MAX = 4096;
int d2h_sync(void* host, const void* dev, size_t bytes, hipStream_t s);
// peer access between the devices of two contexts: 1 = enabled (or same device), 0 = the runtime refused (copies then stage
// through the host inside hipMemcpyPeerAsync; still correct)
int ctx_peer_enabled(int a, int b);
// One host thread at a time (include/mzk.h): every entry point that touches contexts, workspaces or caches starts with
// MZK_ENTER().  A second THREAD arriving while a call is in progress gets MZK_E_BUSY before any state is read; nested calls on
// the owning thread (entry points calling entry points, callbacks calling back in) pass.
struct EntryGuard {
  bool ok;
  bool nested;      // this thread was already inside a call (an entry point calling an entry point, a callback calling back in)
  EntryGuard();
  ~EntryGuard();
};
#define MZK_ENTER()                        \
  mzk::EntryGuard _mzk_entry;              \
  if (!_mzk_entry.ok) return MZK_E_BUSY;   \
  MZK_TRY(mzk::ensure_init())
Context& ctx();
int ctx_count();
int ctx_select(int index);       // hipSetDevice + make it current
int ensure_init();
// Makes context `index` current for a scope and restores the previous context AND the caller's HIP device on exit
// (torch keeps its own idea of the current device).
struct CtxScope {
  int prev_ctx, prev_dev; bool ok;
  explicit CtxScope(int index);
  ~CtxScope();
};

// Grow-only device scratch buffers of the current context, keyed by slot, so steady-state calls never hipMalloc.
int ws_get(WsSlot slot, size_t bytes, void** out);
void ws_release_all();
// Frees the slots of the current context that the running outermost call has not asked for (all of them between calls); waits for
// the device first (earlier *_dev calls may still be reading them).  Returns the bytes given back.
size_t ws_trim_idle();
size_t ws_bytes_held();
// hipMalloc that tries again after ws_trim_idle() when the device is out of memory; MZK_E_NOMEM if it still is.
int dev_alloc(void** out, size_t bytes, const char* what);
size_t poly_bytes_held();                      // mzk_poly.hip: pool blocks + cached interpolation plans of the current context
size_t poly_trim_idle();                       // ... released but for what the running call uses; bytes given back
uint64_t ws_generation();
// Put one at the top of every entry point that enqueues work using workspace slots on stream s.
struct WsGuard {
  hipStream_t s;
  explicit WsGuard(hipStream_t s_);
  ~WsGuard();
};

// ---- per-phase timing (no-ops unless mzk_prof_enable(1)) ---------------------------------------------
void prof_begin(hipStream_t s, int phase);
void prof_end(hipStream_t s, int phase);
struct ProfScope {
  hipStream_t s; int ph;
  ProfScope(hipStream_t s_, int ph_) : s(s_), ph(ph_) { prof_begin(s, ph); }
  ~ProfScope() { prof_end(s, ph); }
};

// ---- host parameter math (O(log n) scalar work: roots, n^-1, canonical checks; never on the data path)
struct HostField {
  int nl;            // u64 limbs
  uint64_t p[4];
};
const HostField* host_field(int fid);
bool h_is_canonical(const HostField* f, const uint64_t* a);
void h_mulmod(const HostField* f, uint64_t* r, const uint64_t* a, const uint64_t* b);
void h_powmod_u64(const HostField* f, uint64_t* r, const uint64_t* a, uint64_t e);
void h_invmod(const HostField* f, uint64_t* r, const uint64_t* a);   // a^(p-2) on the host (parameters only)
void h_ninv_pow2(const HostField* f, unsigned log2n, uint64_t* out);  // (2^log2n)^-1 mod p, needs 2^log2n | p-1
bool h_is_one(const HostField* f, const uint64_t* a);

static inline int field_words(int fid) { return fid == MZK_FIELD_M128 ? 4 : 8; }   // u32 words per element
static inline int field_limbs64(int fid) { return fid == MZK_FIELD_M128 ? 2 : 4; }
static inline size_t field_bytes(int fid) { return fid == MZK_FIELD_M128 ? 16 : 32; }

// ---- entry points implemented per translation unit ------------------------------------------------
int ntt_dev_impl(int fid, const uint64_t* root_host, const void* d_in, void* d_out, size_t n, int inverse,
                 const uint64_t* extra_scale_host, hipStream_t s);
int ntt_batch_dev_impl(int fid, const uint64_t* root_host, const void* d_in, void* d_out, size_t n, size_t batch, int inverse, hipStream_t s);
int coset_lde_dev_impl(int fid, const void* d_coef, size_t n_coef, const uint64_t* offset_host,
                       const uint64_t* generator_host, void* d_out, size_t order, hipStream_t s, size_t batch = 1);
int ntt_columns_dev_impl(int fid, const uint64_t* root_host, const void* d_in, void* d_out, size_t n_points, size_t cols, int inverse, hipStream_t s);
int transpose_elems_dev_impl(int fid, const void* d_in, void* d_out, size_t rows, size_t cols, hipStream_t s);
int poly_scale_dev_impl(int fid, const void* d_in, size_t n, const uint64_t* ratio_host, const uint64_t* lead_host, void* d_out, hipStream_t s);
int coset_divide_dev_impl(int fid, const void* d_lhs, size_t tl, const void* d_rhs, size_t tr, const uint64_t* offset_host,
                          const uint64_t* root_host, size_t order, void* d_out, hipStream_t s);
int pointwise_div_dev(int fid, const void* d_a, const void* d_b, void* d_out, size_t n, hipStream_t s);
int pointwise_div_shared_dev(int fid, const void* d_a, size_t a_stride, const void* d_b, void* d_out, size_t out_stride, size_t n, size_t regs, hipStream_t s);
int pointwise_mul_shared_dev(int fid, const void* d_a, size_t a_stride, const void* d_b, void* d_out, size_t out_stride, size_t n, size_t regs, hipStream_t s);
int pointwise_mul_dev(int fid, const void* d_a, const void* d_b, void* d_out, size_t n, hipStream_t s);
void ntt_release_plans();
void kzg_release_cache();       // mzk_kzg.hip: fixed-base tables of the current context
void poly_release_pool();       // mzk_poly.hip: parked scratch blocks of the current context
int fri_fold_dev_impl(int fid, const void* d_cw, size_t n, const uint64_t* alpha, const uint64_t* offset, const uint64_t* omega,
                      void* d_out, hipStream_t s);
struct FriFoldConsts { uint64_t half[4], oinv[4], winv[4], rmod[4]; };
int fri_fold_consts(int fid, const uint64_t* offset, const uint64_t* omega, FriFoldConsts* fc);
void fri_fold_consts_square(int fid, FriFoldConsts* fc);
int fri_fold_dev_consts(int fid, const void* d_cw, size_t n, const uint64_t* alpha, const FriFoldConsts& fc, void* d_out, hipStream_t s);
int kzg_batch_open_dev(const void* d_coef, size_t n, const uint64_t* us_host, size_t k, const void* d_points, int point_kind,
                       size_t table_stride, void* d_ys, void* d_w_xy, hipStream_t s);

enum { MSM_PTS_PLAIN = 0, MSM_PTS_MONT = 1, MSM_PTS_TABLES = 2 };
// Fixed-base window tables: c-bit signed windows, 254 / c + 1 of them.  point_kind carries c in bits 8..15
// (MSM_PTS_TABLES alone = 16).  Width by SRS size, from the sweep of round 3 (tools/timing/window_sweep.py,
// profiles/r03b_window_sweep.txt; one box, ms per commit at c = 16 / 17 / 19 / 20):
//   2^17  0.54 / 0.57 / 0.89 / 0.80      2^20  1.67 / 1.63 / 2.24 / 1.78      2^22   6.46 /  6.12 /  6.98 / 5.85
//   2^18  0.73 / 0.74 / 1.00 / 0.93      2^21  3.28 / 2.94 / 3.83 / 3.20      2^24  24.38 / 23.00 / 25.01 /  --
//   2^19  1.01 / 0.99 / 1.37 / 1.17
// 17 bits = 15 tables (one accumulation fewer per pair, one table less to hold) with 2^16 buckets wins from 2^19 points on;
// wider windows lose what they save in the accumulation to the sort and to the bucket reduction (2^19 buckets at c = 20:
// reduce 0.42 instead of 0.25 ms) except at exactly 2^22.  Widths whose top window is nearly empty are pathological for the
// merged layout: 254 = 14 * 18 + 2 = 18 * 14 + 2, so at c = 18 (and 14, 21) a quarter of all top-window digits land in each of
// four buckets, 4096 segment partials apiece at 2^20 -- 0.45 ms of heavy-bucket combine at every size.
// Other widths stay selectable through mzk_srs_from_device_ex for tuning and tests (BASELINE configs[2] names 16 bits:
// bench.py reports that width as its own leg).
static inline int msm_table_windows(int c) { return 254 / c + 1; }
// tables a handle with `sets` bucket sets holds: every sets-th window, ceil(windows / sets) = 254 / (c sets) + 1
static inline int msm_table_rows(int c, int sets) { return 254 / (c * sets) + 1; }
// Small SRS (the reference's actual sizes: a few thousand powers at most) take the short paths of mzk_msm.hip (two launches:
// k_small_accumulate_scan + the tail); with tables there is no window Horner either (its ~120 serial doublings are the latency
// floor of a small generic MSM), so they get narrow windows: 8 bits = 32 tables x 128 buckets up to 1024 points, 10 bits =
// 26 tables x 512 buckets up to 2^14 (where the sortless path still beats the general pipeline: profiles/r03v_*), then 16 and,
// from 2^19 points on, 17 bits, from 2^22 on 20 bits through the general pipeline (tools/timing/window_sweep.py).
// Round 4 (tools/gpu_jobs/r04_window_sweep.sh, profiles/round4_window_sweep.txt; one box, c = 16 / 17 / 20 / 22):
//   2^22   6.09 /  5.75 /  5.62 / 12.97      2^23  11.97 / 11.27 / 10.89 / 20.26      2^24  23.56 / 22.05 / 20.89 / 30.18
// 20 bits (13 tables, 2^19 buckets) win from 2^22 points on: two accumulations fewer per pair outweigh the wider sort and the
// longer reduction; 22 bits (12 tables) lose everything to the sort (2^21 buckets: 8192 per coarse bin, past the staged fine
// scatter) and their accumulate is no faster (a longer bucket search per segment).
static inline int msm_srs_window_bits(size_t n) {
  return n <= 1024 ? 8 : (n <= ((size_t)1 << 14) ? 10 : (n < ((size_t)1 << 19) ? 16 : (n < ((size_t)1 << 22) ? 17 : 20)));
}
static inline bool msm_srs_default_tables(size_t n) { return n > 0; }
#define MSM_PTS_TABLES_C(c) (MSM_PTS_TABLES | ((c) << 8))      // bits 16..23: bucket sets (0 or 1 = one)
// One MSM computed in CHUNKS of consecutive pairs (msm_chunked_impl): every chunk is sorted and accumulated on its own -- as soon as
// ITS scalars (and points) are on the device -- into its own bucket array, the chunks' bucket arrays are summed and reduced once.
// msm_dev_impl in chunk mode (cc != null) takes the chunk's scalars, the WHOLE point array, and stops after the segment combine.
struct MsmChunkCtx {
  int k, K;              // this chunk, number of chunks
  size_t i0;             // index of the chunk's first pair in the whole problem
  size_t n_total;        // pairs of the whole problem: decides the window width / bucket layout of every chunk
  size_t n_alloc;        // the largest chunk: size of the per-chunk buffers (the same in every call, so that no workspace slot is regrown)
  hipStream_t sort_stream;   // where the digit sort runs (null: the main stream): a chunk's sort may run under the previous chunk's accumulate
};
struct MsmChunk { const void* d_scalars; size_t i0, n; hipEvent_t ready; };    // ready (or null): recorded when the chunk's inputs are on the device
int msm_chunked_impl(const MsmChunk* chunks, int K, const void* d_points, size_t n_total, int point_kind, size_t table_stride, void* d_out,
                     bool out_partial_xyzz, hipStream_t s, hipStream_t sort_stream, const std::function<int(int)>* before_chunk = nullptr);
bool msm_chunkable(size_t n_total, int point_kind);      // large enough, and a layout the chunk mode covers
int msm_dev_impl(const void* d_scalars, const void* d_points, size_t n, int point_kind, size_t table_stride, void* d_out,
                 bool out_partial_xyzz, hipStream_t s, const std::function<int()>* points_ready = nullptr, const MsmChunkCtx* cc = nullptr);

// Grid-batched form for many short polynomials against one set of narrow window tables (mzk_msm.hip, mzk_kzg.hip)
bool msm_many_supported(int window_bits);
int msm_generic_window_bits(size_t n);        // window width of the generic (GLV) layout for n pairs
int msm_many_dev_impl(const void* d_scalars, size_t n, size_t stride_elems, size_t count, const void* d_tables, int c, size_t table_stride, void* d_out,
                      hipStream_t s);
constexpr size_t MSM_DIRECT_MAX_N = (size_t)1 << 14;
size_t msm_direct_bytes(size_t n, int c);
int msm_build_direct(const void* d_points_mont, size_t n, int c, void* d_direct, hipStream_t s);
int msm_direct_many_dev_impl(const void* d_scalars, size_t n, size_t stride_elems, size_t count, const void* d_direct, int c, size_t table_stride, void* d_out,
                             hipStream_t s);

int xyzz_batch_to_affine(const void* d_xyzz, size_t count, void* d_out, bool out_mont, hipStream_t s);
int msm_build_tables(const void* d_points_mont, size_t n, void* d_tables, int window_bits, hipStream_t s);
int msm_fold_partials_impl(const void* d_partials, int count, void* d_out_xy, hipStream_t s);
int msm_prepare_points(const void* d_points_plain, size_t n, void* d_points_mont, void* d_phi, hipStream_t s);
int msm_phi_points(const void* d_points_mont, size_t n, void* d_phi, hipStream_t s);
int synth_field_impl(int fid, uint64_t seed, size_t n, void* d_out, hipStream_t s);
int synth_g1_impl(uint64_t seed, size_t n, void* d_out, hipStream_t s);
int selftest_inv_wave_impl(uint64_t seed, size_t n, uint64_t* mismatches_host, hipStream_t s);
int selftest_row_ec_impl(uint64_t seed, size_t n, int dbl_reps, uint64_t* mismatches_host, hipStream_t s);
int selftest_field_asm_impl(int fid, uint64_t seed, size_t n, uint64_t* mismatches_host, hipStream_t s);
int selftest_copy_impl(const void* d_src, void* d_dst, size_t bytes, hipStream_t s);
int kzg_setup_g1_dev(const uint64_t* alpha_host, const uint64_t* g1_host, size_t first, size_t count, void* d_powers_xy, hipStream_t s);
int kzg_open_dev(const void* d_coef, size_t n, const uint64_t* u_host, const void* d_points, int point_kind, size_t table_stride,
                 void* d_y, void* d_w_xy, void* d_q_out, hipStream_t s, bool value_only = false);

}  // namespace mzk

// Device-resident PublicKeyKZG.powers_1 (kzg.rs:8-11)
struct mzk_srs {
  void* d_points_mont;   // msm_table_windows(window_bits) x n window tables when has_tables, else n prepared points + their n phi images
  size_t n;
  bool has_tables;
  int window_bits;
  int ctx_index;         // the context (device) that owns d_points_mont
  // Bucket sets of the table layout (1 = every window shares one set: the default).  With k sets the handle holds only every
  // k-th window table, T[q][i] = 2^(c k q) P_i: window w = k q + r adds T[q][i] into bucket set r, and the result is
  // sum_r 2^(c r) B_r (k - 1 times c doublings at the very end).  1/k of the table memory for the same additions -- what a
  // handle degrades to when the full tables do not fit the budget / the device (srs_alloc_layout).
  int sets = 1;
  // optional (mzk_srs_build_direct): every multiple a window digit can ask for, D[(w n + i) 2^(direct_bits-1) + m] = (m + 1) 2^(direct_bits w) P_i,
  // affine Montgomery -- commitments of many short polynomials then need no buckets at all (msm_many_srs)
  void* d_direct = nullptr;
  int direct_bits = 0;
  size_t direct_bytes = 0;
  // optional, built by the first grid-batched pass that wants them (msm_many_srs): a SECOND set of window tables, wider than the
  // handle's own.  A handle of <= 2^14 points keeps 8- / 10-bit tables for the sortless single commit; a batch of LONG polynomials
  // (>= 2^13 coefficients) is 20 % faster over 12-bit ones (shorter accumulate, and a bucket's partials are a chain a quarter as
  // long in the segment combine: profiles/round5_many_commit_widths.txt).  22 x n points: 22 MiB at 2^14.
  mutable void* d_tables_wide = nullptr;
  mutable int wide_bits = 0;
  mutable size_t wide_bytes = 0;
  int kind() const { return has_tables ? (mzk::MSM_PTS_TABLES | (window_bits << 8) | (sets << 16)) : (int)mzk::MSM_PTS_MONT; }
  size_t table_rows() const { return has_tables ? (size_t)mzk::msm_table_rows(window_bits, sets) : 2; }   // no tables: P_i, then phi(P_i) (GLV layout)
};

namespace mzk {
// `count` commitments of n coefficients each (polynomial j at d_scalars + j * stride_elems * 32 bytes) against one handle, as ONE
// pass: over the direct tables when the handle has them, over its narrow window tables otherwise (srs_many_capable)
bool srs_many_capable(const mzk_srs* srs);
int msm_many_srs(const mzk_srs* srs, const void* d_scalars, size_t n, size_t stride_elems, size_t count, void* d_out, hipStream_t s);
bool kzg_open_many_supported(const mzk_srs* srs, size_t n);
int kzg_open_many_dev(const mzk_srs* srs, const void* d_coefs, size_t n, size_t count, const uint64_t* us_host, void* d_ys, void* d_ws_xy, hipStream_t s);
}  // namespace mzk
