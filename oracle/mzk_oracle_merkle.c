/* CPU ORACLE (test infrastructure, never shipped, never measured as the product) -- Merkle / SHA3 / leaf bytes.
 *
 * Restates, for SURVEY.md section 8(f) rank 1:
 *   - Merkle::hash / commit / open / verify           merkle.rs:8-13, 15-25, 28-46, 49-67
 *   - the leaf bytes FRI and the STARK provers feed it fri.rs:160-166 (bincode::serialize of each
 *     FiniteFieldElement, field.rs:87-91)
 * SHA3-256 is FIPS 202 (the reference uses the `sha3` crate 0.10.8, myzkp/Cargo.toml:14; not vendored): pinned
 * here by the NIST known answers and, in tests/, against Python's hashlib on random inputs.
 *
 * PARITY UNPINNED at the leaf-byte layer: the byte layout of bincode(FiniteFieldElement) is decided by
 * num-bigint 0.4's serde impl and bincode 1.3.3 (myzkp/Cargo.toml:7,15), neither of which is under
 * /root/reference, and no reference test holds a golden root.  The layout restated below is the published one:
 *   BigInt      -> tuple (Sign, BigUint)
 *   Sign        -> i8: Minus = -1, NoSign = 0, Plus = 1
 *   BigUint     -> sequence of u32 digits, little-endian, no leading (most significant) zero digit; zero -> empty
 *   bincode 1.x -> fixed-width little-endian ints, sequence length as u64, PhantomData -> no bytes
 * (restated from the crates' published sources as remembered, none of them available offline here:
 *  num-bigint 0.4 src/bigint/serde.rs -- `impl Serialize for Sign` writes -1i8 / 0i8 / 1i8 and `impl Serialize for
 *  BigInt` the tuple (sign, magnitude); src/biguint/serde.rs -- "always serialize as a u32 sequence", on 64-bit
 *  digits as lo, hi halves with the top zero half dropped; bincode 1.3.3 `bincode::serialize` =
 *  DefaultOptions + fixint encoding, little endian: seq length u64, tuples/structs without framing.)
 * so a canonical element v > 0 with k significant u32 digits is  01 | k as u64 LE | k x u32 LE  (9 + 4k bytes)
 * and v = 0 is  00 | 0 as u64 LE  (9 bytes).  Elements are canonical here (fri.rs:190 sanitizes each fold; the
 * shim sanitizes the initial codeword), so Sign::Minus never occurs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint8_t u8;

/* ---- Keccak-f[1600] / SHA3-256 (FIPS 202) ---------------------------------------------------------- */
static const u64 KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static u64 rotl64(u64 x, int r) { return r ? (x << r) | (x >> (64 - r)) : x; }
static void keccak_f(u64 a[25]) {
  for (int rnd = 0; rnd < 24; rnd++) {
    u64 c[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) {
      u64 d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
      for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
    }
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], KECCAK_ROT[x + 5 * y]);
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= KECCAK_RC[rnd];
  }
}
void orc_sha3_256(const u8* data, size_t len, u8 out[32]) {
  enum { RATE = 136 };
  u64 st[25];
  memset(st, 0, sizeof st);
  u8 block[RATE];
  while (len >= RATE) {
    for (int i = 0; i < RATE / 8; i++) { u64 w; memcpy(&w, data + 8 * i, 8); st[i] ^= w; }
    keccak_f(st);
    data += RATE; len -= RATE;
  }
  memset(block, 0, RATE);
  memcpy(block, data, len);
  block[len] ^= 0x06;
  block[RATE - 1] ^= 0x80;
  for (int i = 0; i < RATE / 8; i++) { u64 w; memcpy(&w, block + 8 * i, 8); st[i] ^= w; }
  keccak_f(st);
  memcpy(out, st, 32);
}

/* ---- bincode(FiniteFieldElement) of a canonical element given as nl u64 limbs -------------------------- */
size_t orc_bincode_field(const u64* limbs, int nl, u8* out /* >= 9 + 8 nl */) {
  int k = 2 * nl;
  while (k > 0) {
    uint32_t d = (uint32_t)(limbs[(k - 1) / 2] >> (32 * ((k - 1) & 1)));
    if (d) break;
    k--;
  }
  out[0] = k ? 1 : 0;
  u64 len = (u64)k;
  memcpy(out + 1, &len, 8);
  for (int i = 0; i < k; i++) {
    uint32_t d = (uint32_t)(limbs[i / 2] >> (32 * (i & 1)));
    memcpy(out + 9 + 4 * i, &d, 4);
  }
  return 9 + 4 * (size_t)k;
}

/* bincode(FiniteFieldElement) of an element whose BigInt the reference left negative (field.rs:98-110: `%` keeps the
 * sign; only sanitize() removes it): Sign::Minus as i8 = 0xff, then the MAGNITUDE's digits.  -0 does not exist. */
size_t orc_bincode_field_signed(const u64* magnitude, int nl, int negative, u8* out) {
  size_t l = orc_bincode_field(magnitude, nl, out);
  if (negative && out[0]) out[0] = 0xff;
  return l;
}

/* ---- Merkle over byte leaves: leaves[offsets[i] .. offsets[i+1]) ----------------------------------------- */
/* merkle.rs:15-25 -- a single leaf commits to ITSELF (unhashed); otherwise hash(commit(left) || commit(right)). */
static size_t merkle_commit_rec(const u8* leaves, const u64* off, size_t lo, size_t cnt, u8* out /* >= max(32, leaf) */) {
  if (cnt == 1) {
    size_t l = (size_t)(off[lo + 1] - off[lo]);
    memcpy(out, leaves + off[lo], l);
    return l;
  }
  size_t mid = cnt / 2;
  size_t cap = 64;
  for (size_t i = lo; i < lo + cnt; i++) { size_t l = (size_t)(off[i + 1] - off[i]); if (2 * l > cap) cap = 2 * l; }
  u8* buf = (u8*)malloc(cap);
  size_t a = merkle_commit_rec(leaves, off, lo, mid, buf);
  size_t b = merkle_commit_rec(leaves, off, lo + mid, cnt - mid, buf + a);
  orc_sha3_256(buf, a + b, out);
  free(buf);
  return 32;
}
static size_t max_leaf(const u64* off, size_t n) {
  size_t m = 32;
  for (size_t i = 0; i < n; i++) if ((size_t)(off[i + 1] - off[i]) > m) m = (size_t)(off[i + 1] - off[i]);
  return m;
}
/* root buffer must hold max(32, longest leaf) bytes */
int orc_merkle_commit_ref(const u8* leaves, const u64* offsets, size_t n, u8* root, size_t* root_len) {
  if (n == 0) return -1;                       /* merkle.rs:17-22 would recurse forever on an empty slice */
  *root_len = merkle_commit_rec(leaves, offsets, 0, n, root);
  return 0;
}
/* merkle.rs:28-46 -- path entries bottom-up; entry k goes to path + path_off[k], its length to path_len[k]. */
static size_t merkle_open_rec(const u8* leaves, const u64* off, size_t lo, size_t cnt, size_t index, u8* path, u64* plen, size_t depth_done,
                              size_t stride) {
  if (cnt == 2) {
    size_t sib = lo + (1 - index);
    size_t l = (size_t)(off[sib + 1] - off[sib]);
    memcpy(path + depth_done * stride, leaves + off[sib], l);
    plen[depth_done] = l;
    return depth_done + 1;
  }
  size_t mid = cnt / 2;
  size_t d;
  if (index < mid) {
    d = merkle_open_rec(leaves, off, lo, mid, index, path, plen, depth_done, stride);
    plen[d] = merkle_commit_rec(leaves, off, lo + mid, cnt - mid, path + d * stride);
  } else {
    d = merkle_open_rec(leaves, off, lo + mid, cnt - mid, index - mid, path, plen, depth_done, stride);
    plen[d] = merkle_commit_rec(leaves, off, lo, mid, path + d * stride);
  }
  return d + 1;
}
/* path: depth entries of `stride` bytes each (stride >= max(32, longest leaf)); returns the depth in *depth */
int orc_merkle_open_ref(const u8* leaves, const u64* offsets, size_t n, size_t index, u8* path, u64* path_len, size_t stride, size_t* depth) {
  if (n < 2 || index >= n || stride < max_leaf(offsets, n)) return -1;
  /* merkle.rs:32-45 terminates only when the descent reaches a two-leaf slice; a one-leaf slice has mid = 0 and
   * recurses on itself forever (stack overflow in the reference): report that instead of reproducing it. */
  for (size_t cnt = n, idx = index; cnt != 2;) {
    if (cnt == 1) return -2;
    size_t mid = cnt / 2;
    if (idx < mid) cnt = mid; else { idx -= mid; cnt -= mid; }
  }
  *depth = merkle_open_rec(leaves, offsets, 0, n, index, path, path_len, 0, stride);
  return 0;
}
/* merkle.rs:49-67 */
int orc_merkle_verify_ref(const u8* root, size_t root_len, size_t index, const u8* path, const u64* path_len, size_t stride, size_t depth,
                          const u8* leaf, size_t leaf_len) {
  if (depth == 0) return 0;
  size_t cap = leaf_len + 64;
  for (size_t k = 0; k < depth; k++) cap += (size_t)path_len[k];
  u8* cur = (u8*)malloc(cap);
  u8* buf = (u8*)malloc(cap + 64);
  size_t cl = leaf_len;
  memcpy(cur, leaf, leaf_len);
  for (size_t k = 0; k < depth; k++) {
    const u8* sib = path + k * stride;
    size_t sl = (size_t)path_len[k];
    if (index % 2 == 0) { memcpy(buf, cur, cl); memcpy(buf + cl, sib, sl); }
    else { memcpy(buf, sib, sl); memcpy(buf + sl, cur, cl); }
    orc_sha3_256(buf, cl + sl, cur);
    cl = 32;
    index >>= 1;
  }
  int ok = (root_len == 32) && memcmp(root, cur, 32) == 0;
  free(cur); free(buf);
  return ok;
}
/* convenience for the tests: serialize n field elements into leaves + offsets (offsets has n+1 entries) */
void orc_bincode_field_vector(const u64* elems, int nl, size_t n, u8* leaves, u64* offsets) {
  u64 o = 0;
  for (size_t i = 0; i < n; i++) {
    offsets[i] = o;
    o += orc_bincode_field(elems + (size_t)nl * i, nl, leaves + o);
  }
  offsets[n] = o;
}
