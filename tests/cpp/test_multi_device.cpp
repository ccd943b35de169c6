// BASELINE configs[3] through the C ABI alone -- no Python, no torch: one process, W contexts (mzk_init_devices),
// contiguous shards, gather of the 128-byte XYZZ partials, fold.  world runs 1..8 over however many devices are
// visible (ordinals wrap around, so a one-GPU box exercises 8 contexts on device 0); when more than one device is
// visible the first pass uses each device once.
//   KZG commit of 2^16 + 5 coefficients against an SRS built shard-wise on the GPUs (setup_kzg, kzg.rs:27-40):
//       commit == [f(alpha)] G  (polynomial.rs:156-165, kzg.rs:57-59)
//   generic MSM over host arrays == the oracle's Pippenger
//   one 2^14-point Fr transform spread over the contexts (mzk_ntt_multi, worlds 1, 2, 4, 8) == the oracle's transform
// Usage: test_multi_device [n_visible_devices]   (the count is passed in by the Python harness; default 1)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/mzk.h"

extern "C" {  // oracle (checker only)
int orc_poly_eval(int fid, const uint64_t* coef, size_t n, const uint64_t* x, uint64_t* out);
int orc_ec_mul(int cid, const uint64_t* p_xy, const uint64_t* k, int nk, uint64_t* out_xy);
int orc_msm_fast(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t* out_xy, int nthreads);
void orc_synth_vector(int fid, uint64_t seed, size_t n, uint64_t* out, int nthreads);
void orc_synth_g1_points(uint64_t seed, size_t n, uint64_t* out, int nthreads);
int orc_ntt_fast(int fid, const uint64_t* root, const uint64_t* in, uint64_t* out, size_t n, int inverse, int nthreads);
}
static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s  (%s)\n", __FILE__, __LINE__, #c, mzk_last_error()); failures++; } } while (0)
static bool same(const uint64_t* a, const uint64_t* b, int n) { for (int i = 0; i < n; i++) if (a[i] != b[i]) return false; return true; }

int main(int argc, char** argv) {
  const int visible = argc > 1 ? atoi(argv[1]) : 1;
  const size_t n = (1u << 16) + 5, m = 3001;
  const uint64_t g1[8] = {1, 0, 0, 0, 2, 0, 0, 0};
  uint64_t alpha[4];
  orc_synth_vector(MZK_FIELD_FR, 31337, 1, alpha, 1);
  std::vector<uint64_t> coef(n * 4), sc(m * 4), pts(m * 8);
  orc_synth_vector(MZK_FIELD_FR, 31338, n, coef.data(), 8);
  orc_synth_vector(MZK_FIELD_FR, 31339, m, sc.data(), 8);
  orc_synth_g1_points(31340, m, pts.data(), 8);
  uint64_t fa[4], want_commit[8], want_msm[8];
  CHECK(orc_poly_eval(MZK_FIELD_FR, coef.data(), n, alpha, fa) == 0);
  CHECK(orc_ec_mul(0, g1, fa, 4, want_commit) == 0);
  CHECK(orc_msm_fast(sc.data(), pts.data(), m, want_msm, 8) == 0);
  const size_t nt = 1u << 14;
  uint64_t wt[4];
  CHECK(mzk_root_of_unity(MZK_FIELD_FR, 14, wt) == MZK_OK);
  std::vector<uint64_t> xt(4 * nt), want_ntt(4 * nt), want_intt(4 * nt);
  orc_synth_vector(MZK_FIELD_FR, 31341, nt, xt.data(), 8);
  CHECK(orc_ntt_fast(MZK_FIELD_FR, wt, xt.data(), want_ntt.data(), nt, 0, 8) == 0);
  CHECK(orc_ntt_fast(MZK_FIELD_FR, wt, xt.data(), want_intt.data(), nt, 1, 8) == 0);
  for (int world = 1; world <= 8; world++) {
    int ord[8];
    for (int r = 0; r < world; r++) ord[r] = r % (visible > 0 ? visible : 1);
    CHECK(mzk_init_devices(ord, world) == MZK_OK);
    CHECK(mzk_ctx_count() == world);
    for (int with_tables = 0; with_tables <= 1; with_tables++) {
      mzk_srs_multi* h = nullptr;
      CHECK(mzk_kzg_setup_srs_multi(alpha, g1, n - 1, with_tables, &h) == MZK_OK);
      if (!h) continue;
      CHECK(mzk_srs_multi_world(h) == world && mzk_srs_multi_shard_lo(h, world) == n);
      uint64_t got[8];
      CHECK(mzk_kzg_commit_srs_multi(h, coef.data(), n, got) == MZK_OK);
      CHECK(same(got, want_commit, 8));
      mzk_srs_multi_free(h);
    }
    uint64_t got[8];
    CHECK(mzk_msm_g1_bn254_multi(sc.data(), pts.data(), m, got) == MZK_OK);
    CHECK(same(got, want_msm, 8));
    // one 2^14-point transform spread over the contexts (power-of-two worlds): mzk_ntt_multi == the oracle's transform
    if ((world & (world - 1)) == 0) {
      for (int inverse = 0; inverse <= 1; inverse++) {
        std::vector<uint64_t> y(4 * nt);
        CHECK(mzk_ntt_multi(MZK_FIELD_FR, wt, xt.data(), y.data(), nt, inverse) == MZK_OK);
        CHECK(y == (inverse ? want_intt : want_ntt));
      }
    }
    printf("world %d (devices:", world);
    for (int r = 0; r < world; r++) printf(" %d", mzk_ctx_device(r));
    printf(") ok\n");
  }
  mzk_shutdown();
  if (failures) { printf("%d failures\n", failures); return 1; }
  printf("multi-device C ABI test passed\n");
  return 0;
}
