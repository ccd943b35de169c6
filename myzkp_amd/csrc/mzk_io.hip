// mzk_io.hip -- persistence of a device-resident SRS (SURVEY 8f rank 3 "SRS dump for PublicKeyKZG", section 5 "checkpoint").
//
// The reference keeps PublicKeyKZG only in memory (kzg.rs:8-11; its types derive serde but nothing writes them), so
// the format is this library's: a 64-byte header, then the n affine points exactly as they cross the C ABI --
// x || y, 4 + 4 little-endian u64 limbs, canonical, all-zero = infinity (64 n bytes) -- then, optionally, the
// window tables 2^(c w) P_i for w >= 1 in the library's internal encoding (Montgomery words, 29-bit-limb radix; tied
// to `format`).  The header carries an FNV-1a 64 of the point bytes and (format 2) a second checksum over the table
// section, seeded with the first so that tables are keyed to their points: a load that does not reproduce BOTH fails with
// MZK_E_IO (truncated, corrupted or stale file) -- a flipped bit in a table row would otherwise turn every later commit
// into a wrong group element returned with MZK_OK.  Format-1 files (points checksum only) still load; their table
// section is ignored and the tables are rebuilt from the checked points.
//
// Loading tables from disk is a convenience, not a speed-up: building them takes ~17 ms per 2^20 points on the GPU,
// less than reading their 1 GiB from any disk -- save with_tables = 0 unless the consumer cannot afford the build's
// transient memory.
#include <stdio.h>
#include <vector>
#include "mzk_common.h"

using namespace mzk;

namespace mzk {
int srs_alloc_layout(mzk_srs* h, int with_tables);     // mzk_api.hip
struct SrsFileHeader {
  char magic[8];          // "MZKSRS\0\0"
  uint32_t format;        // 2 (1 = older dumps without tables_fnv1a)
  uint32_t flags;         // bit 0: window tables follow the points
  uint64_t n;
  uint32_t window_bits;   // of the stored tables (0 if none)
  uint32_t table_rows;    // rows stored after the points: msm_table_windows(window_bits) - 1
  uint64_t points_fnv1a;
  uint64_t tables_fnv1a;  // format 2: word-wise FNV-1a of the table section, seeded with points_fnv1a (0 if no tables)
  uint64_t reserved[2];
};
static_assert(sizeof(SrsFileHeader) == 64, "header layout");
static const char SRS_MAGIC[8] = {'M', 'Z', 'K', 'S', 'R', 'S', 0, 0};
static const size_t IO_CHUNK_POINTS = (size_t)1 << 19;   // 32 MiB per transfer

static uint64_t fnv1a(uint64_t h, const void* data, size_t len) {
  const uint8_t* p = (const uint8_t*)data;
  for (size_t i = 0; i < len; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
  return h;
}
// the table section is up to 15 GiB: 8 bytes per step (len is a multiple of 64)
static uint64_t fnv1a_words(uint64_t h, const void* data, size_t len) {
  const uint64_t* p = (const uint64_t*)data;
  for (size_t i = 0; i < len / 8; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
  return h;
}
int msm_points_to_plain(const void* d_points_mont, size_t n, void* d_points_plain, hipStream_t s);   // mzk_msm.hip
struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
}  // namespace mzk

extern "C" {

int mzk_srs_save(const mzk_srs* srs, const char* path, int with_tables) {
  MZK_ENTER();
  if (!srs || !path) { set_error("srs_save: null pointer"); return MZK_E_ARG; }
  if (srs->ctx_index != ctx().index) { set_error("SRS handle lives on context %d, the current context is %d", srs->ctx_index, ctx().index); return MZK_E_ARG; }
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  FileCloser fc{fopen(path, "wb")};
  if (!fc.f) { set_error("srs_save: cannot open %s for writing", path); return MZK_E_IO; }
  const bool tables = with_tables && srs->has_tables && srs->sets == 1;      // a degraded layout (every k-th table) is rebuilt from the points on load
  SrsFileHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, SRS_MAGIC, 8);
  h.format = 2; h.flags = tables ? 1u : 0u; h.n = srs->n;
  h.window_bits = tables ? (uint32_t)srs->window_bits : 0u;
  h.table_rows = tables ? (uint32_t)(msm_table_windows(srs->window_bits) - 1) : 0u;
  if (fwrite(&h, sizeof h, 1, fc.f) != 1) { set_error("srs_save: write failed"); return MZK_E_IO; }
  std::vector<uint8_t> host(IO_CHUNK_POINTS * 64);
  void* d_plain;
  MZK_TRY(ws_get(WS_MISC_A, IO_CHUNK_POINTS * 64, &d_plain));
  uint64_t hash = 0xcbf29ce484222325ULL;
  for (size_t at = 0; at < srs->n; at += IO_CHUNK_POINTS) {
    const size_t m = srs->n - at < IO_CHUNK_POINTS ? srs->n - at : IO_CHUNK_POINTS;
    MZK_TRY(msm_points_to_plain((const uint8_t*)srs->d_points_mont + at * 64, m, d_plain, s));    // row 0 / the prepared points
    MZK_HIP(hipMemcpyAsync(host.data(), d_plain, m * 64, hipMemcpyDeviceToHost, s));
    MZK_HIP(hipStreamSynchronize(s));
    hash = fnv1a(hash, host.data(), m * 64);
    if (fwrite(host.data(), 64, m, fc.f) != m) { set_error("srs_save: write failed"); return MZK_E_IO; }
  }
  uint64_t thash = hash;
  for (uint32_t row = 1; tables && row <= h.table_rows; row++) {
    for (size_t at = 0; at < srs->n; at += IO_CHUNK_POINTS) {
      const size_t m = srs->n - at < IO_CHUNK_POINTS ? srs->n - at : IO_CHUNK_POINTS;
      MZK_HIP(hipMemcpyAsync(host.data(), (const uint8_t*)srs->d_points_mont + ((size_t)row * srs->n + at) * 64, m * 64, hipMemcpyDeviceToHost, s));
      MZK_HIP(hipStreamSynchronize(s));
      thash = fnv1a_words(thash, host.data(), m * 64);
      if (fwrite(host.data(), 64, m, fc.f) != m) { set_error("srs_save: write failed"); return MZK_E_IO; }
    }
  }
  h.points_fnv1a = hash;
  h.tables_fnv1a = tables ? thash : 0;
  if (fseek(fc.f, 0, SEEK_SET) != 0 || fwrite(&h, sizeof h, 1, fc.f) != 1 || fflush(fc.f) != 0) { set_error("srs_save: write failed"); return MZK_E_IO; }
  return MZK_OK;
}

// with_tables as in mzk_srs_from_device_ex (0 = plain prepared points, 1 = the default width for n (msm_srs_window_bits), 8..22 = that width);
// tables stored in the file are used when their width matches, otherwise they are rebuilt from the points.
int mzk_srs_load(const char* path, int with_tables, mzk_srs** out) {
  MZK_ENTER();
  if (!path || !out) { set_error("srs_load: null pointer"); return MZK_E_ARG; }
  *out = nullptr;
  if (with_tables < 0 || (with_tables > 1 && (with_tables < 8 || with_tables > 22))) { set_error("srs_load: with_tables must be 0, 1 or a window width 8..22"); return MZK_E_ARG; }
  hipStream_t s = ctx().stream;
  FileCloser fc{fopen(path, "rb")};
  if (!fc.f) { set_error("srs_load: cannot open %s", path); return MZK_E_IO; }
  SrsFileHeader h;
  if (fread(&h, sizeof h, 1, fc.f) != 1 || memcmp(h.magic, SRS_MAGIC, 8) != 0 || (h.format != 1 && h.format != 2)) { set_error("srs_load: %s is not an SRS dump of this library", path); return MZK_E_IO; }
  if (h.n > ((uint64_t)1 << 27)) { set_error("srs_load: %llu points exceed the supported 2^27", (unsigned long long)h.n); return MZK_E_IO; }
  const size_t n = (size_t)h.n;
  void* d_plain = nullptr;
  if (hipMalloc(&d_plain, n ? n * 64 : 64) != hipSuccess) { set_error("srs_load: hipMalloc failed"); return MZK_E_HIP; }
  struct DevFree { void* p; ~DevFree() { if (p) (void)hipFree(p); } } df{d_plain};
  std::vector<uint8_t> host(IO_CHUNK_POINTS * 64);
  uint64_t hash = 0xcbf29ce484222325ULL;
  for (size_t at = 0; at < n; at += IO_CHUNK_POINTS) {
    const size_t m = n - at < IO_CHUNK_POINTS ? n - at : IO_CHUNK_POINTS;
    if (fread(host.data(), 64, m, fc.f) != m) { set_error("srs_load: %s is truncated", path); return MZK_E_IO; }
    hash = fnv1a(hash, host.data(), m * 64);
    MZK_HIP(hipMemcpyAsync((uint8_t*)d_plain + at * 64, host.data(), m * 64, hipMemcpyHostToDevice, s));
    MZK_HIP(hipStreamSynchronize(s));     // `host` is reused
  }
  if (hash != h.points_fnv1a) { set_error("srs_load: checksum mismatch in %s (corrupted or truncated)", path); return MZK_E_IO; }
  const int want_bits = with_tables > 1 ? with_tables : msm_srs_window_bits(n);
  const bool use_stored = with_tables && h.format >= 2 && (h.flags & 1u) && (int)h.window_bits == want_bits &&
                          h.table_rows == (uint32_t)(msm_table_windows(want_bits) - 1) && n > 0;
  if (!use_stored) return mzk_srs_from_device_ex(d_plain, n, with_tables, out, s);
  WsGuard wsg(s);
  mzk_srs* hd = new mzk_srs{nullptr, n, false, 0, ctx().index};
  const size_t rows = (size_t)h.table_rows + 1;
  int rc = srs_alloc_layout(hd, with_tables);
  if (rc == MZK_OK && !(hd->has_tables && hd->sets == 1 && hd->window_bits == want_bits)) {
    // the stored tables do not fit the budget / the device as they are: take whatever layout does fit, built from the points
    mzk_srs_free(hd);
    return mzk_srs_from_device_ex(d_plain, n, with_tables, out, s);
  }
  if (rc != MZK_OK) { delete hd; return rc; }
  uint64_t thash = hash;
  if (rc == MZK_OK) rc = msm_prepare_points(d_plain, n, hd->d_points_mont, nullptr, s);      // row 0 = the points, Montgomery form
  for (size_t row = 1; rc == MZK_OK && row < rows; row++) {
    for (size_t at = 0; rc == MZK_OK && at < n; at += IO_CHUNK_POINTS) {
      const size_t m = n - at < IO_CHUNK_POINTS ? n - at : IO_CHUNK_POINTS;
      if (fread(host.data(), 64, m, fc.f) != m) { set_error("srs_load: %s is truncated (tables)", path); rc = MZK_E_IO; break; }
      thash = fnv1a_words(thash, host.data(), m * 64);
      if (hipMemcpyAsync((uint8_t*)hd->d_points_mont + (row * n + at) * 64, host.data(), m * 64, hipMemcpyHostToDevice, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) { set_error("srs_load: copy failed"); rc = MZK_E_HIP; }
    }
  }
  if (rc == MZK_OK && hipStreamSynchronize(s) != hipSuccess) rc = MZK_E_HIP;
  if (rc == MZK_OK && thash != h.tables_fnv1a) { set_error("srs_load: checksum mismatch in the window tables of %s (corrupted or stale)", path); rc = MZK_E_IO; }
  if (rc != MZK_OK) { mzk_srs_free(hd); return rc; }
  *out = hd;
  return MZK_OK;
}

// the points of a handle back at the ABI (affine canonical, n * 8 limbs): PublicKeyKZG.powers_1 as the reference holds it
int mzk_srs_download(const mzk_srs* srs, uint64_t* powers_xy, size_t cap_points) {
  MZK_ENTER();
  if (!srs || (!powers_xy && srs->n)) { set_error("srs_download: null pointer"); return MZK_E_ARG; }
  if (cap_points < srs->n) { set_error("srs_download: buffer holds %zu points, the SRS has %zu", cap_points, srs->n); return MZK_E_LENGTH; }
  if (srs->ctx_index != ctx().index) { set_error("SRS handle lives on context %d, the current context is %d", srs->ctx_index, ctx().index); return MZK_E_ARG; }
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  void* d_plain;
  MZK_TRY(ws_get(WS_MISC_A, IO_CHUNK_POINTS * 64, &d_plain));
  for (size_t at = 0; at < srs->n; at += IO_CHUNK_POINTS) {
    const size_t m = srs->n - at < IO_CHUNK_POINTS ? srs->n - at : IO_CHUNK_POINTS;
    MZK_TRY(msm_points_to_plain((const uint8_t*)srs->d_points_mont + at * 64, m, d_plain, s));
    MZK_HIP(hipMemcpyAsync(powers_xy + at * 8, d_plain, m * 64, hipMemcpyDeviceToHost, s));
    MZK_HIP(hipStreamSynchronize(s));
  }
  return MZK_OK;
}
size_t mzk_srs_len(const mzk_srs* srs) { return srs ? srs->n : 0; }

}  // extern "C"
