/* mzk.h -- C ABI of the MI355X-native MSM / NTT prover path for MyZKP.
 *
 * The reference has no FFI for this path; this header defines the seam.  Each entry
 * point names the reference call site it stands behind (paths relative to myzkp/src/modules/).
 * A Rust maintainer binds these with an `extern "C"` block (INTEGRATION.md shows the shim).
 *
 * Conventions (taken from the reference's only device boundary, examples/sumcheck):
 *   - field elements cross the boundary in STANDARD form as little-endian u64 limbs, canonical in
 *     [0, p): 4 limbs for BN254 Fr/Fq, 2 limbs for M128      (examples/sumcheck/src/utils.rs:51-72)
 *   - an affine G1 point is x||y (8 limbs); the all-zero encoding is the point at infinity
 *     ((0,0) is not on y^2 = x^3 + 3)
 *   - every function returns MZK_OK (0) or a negative MZK_E_* code; mzk_last_error() has the text.
 *     The Rust shim asserts the reference's own preconditions first so that panic messages stay
 *     identical (ntt.rs:8-23, curve.rs:174-176), then `expect()`s the status.
 *   - blocking, one in-flight call per process (the reference is single-threaded); one process per
 *     GPU.  The *_dev variants take device pointers + a hipStream_t and only enqueue work.
 *   - all entry points of a context share one grow-only workspace.  Calls on DIFFERENT streams are ordered by the
 *     library (each entry point waits on the previous one's completion event before touching the workspace and
 *     records its own), so results are correct on any stream; work on one context never overlaps.  Calls must
 *     still come from one host thread at a time: a call that arrives while another thread is inside the library returns
 *     MZK_E_BUSY without touching any state (calls nested on ONE thread -- the challenge callback of mzk_fri_commit calling back
 *     in -- are fine).
 */
#ifndef MZK_H
#define MZK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MZK_FIELD_FR = 0,   /* ModEIP197 / FqOrder: algebra/field.rs:428-431, curve/bn128.rs:30 */
       MZK_FIELD_M128 = 1, /* M128 = 1 + 407*2^119: zkstark/fri.rs:408 */
       MZK_FIELD_FQ = 2    /* BN128Modulus: curve/bn128.rs:19-22 (point coordinates only) */ };

enum { MZK_OK = 0,
       MZK_E_ARG = -1,        /* null pointer / bad field id */
       MZK_E_NOT_POW2 = -2,   /* ntt.rs:8-11 "cannot compute ntt of non-power-of-two sequence" */
       MZK_E_ROOT_ORDER = -3, /* ntt.rs:15-18 "primitive root must be nth root of unity" */
       MZK_E_ROOT_PRIM = -4,  /* ntt.rs:19-22 "primitive root is not primitive nth root of unity" */
       MZK_E_LENGTH = -5,     /* slice-index / usize-underflow panics (ntt.rs:265, polynomial.rs:162) */
       MZK_E_RANGE = -6,      /* operand not canonical (>= modulus) where the ABI requires it */
       MZK_E_HIP = -7,        /* HIP runtime error */
       MZK_E_NOGPU = -8,      /* no gfx950 device visible: there is NO CPU fallback */
       MZK_E_CALLBACK = -9,   /* a caller-supplied callback reported failure (mzk_fri_commit's challenge) */
       MZK_E_IO = -10,        /* file could not be opened / read / written, or is not a valid dump (mzk_srs_save/load) */
       MZK_E_BUSY = -11,      /* another host thread is inside a call: the library serves one call at a time (see above);
                                 nothing was enqueued, the call may simply be repeated */
       MZK_E_NOMEM = -12      /* the device has no memory left for this call even after every idle workspace buffer of the context
                                 was released (see mzk_set_workspace_budget / mzk_trim_workspace); nothing was enqueued */ };

/* Select the device for this process, create streams/workspace.  Idempotent: when context 0 already drives
 * `device_ordinal` nothing is torn down (contexts made by mzk_init_devices, their streams and every handle stay valid);
 * use mzk_init_devices(&ordinal, 1) to go back to exactly one context.  (SURVEY 8b sketched mzk_init(n_devices);
 * one process per GPU is the primary model, so the argument is the ordinal -- mzk_init_devices is the n-device form.) */
int mzk_init(int device_ordinal);
/* One process driving several GPUs: context r = (device_ordinals[r], its own stream, workspace and table caches).
 * Duplicate ordinals are allowed (several contexts on one GPU).  Context 0 is current afterwards; every entry point
 * of this header works on the CURRENT context (mzk_ctx_select), the *_multi entry points walk all of them. */
int mzk_init_devices(const int* device_ordinals, int n_devices);
int mzk_ctx_count(void);
int mzk_ctx_select(int index);
int mzk_ctx_device(int index);   /* device ordinal of a context, -1 if it does not exist */
/* 1 when the devices of contexts a and b can reach each other's memory directly (mzk_init_devices calls
 * hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess for every pair of distinct ordinals; same device = 1), else 0:
 * mzk_ntt_multi's exchanges then stage through host memory inside the runtime -- correct, slower. */
int mzk_ctx_peer_enabled(int index_a, int index_b);
void* mzk_ctx_stream(int index); /* the context's own hipStream_t (non-blocking), usable as the `stream` of its *_dev calls */
/* Two contexts on ONE device keep two calls in flight (each has its own stream and workspace; an mzk_srs handle is
 * plain device memory and serves every context of its device): alternating commits between them overlaps the
 * latency-bound tail of one MSM with the sort / accumulate of the next -- 2^20 pairs: 1.72 -> 1.55 ms per commit. */
void mzk_shutdown(void);
const char* mzk_last_error(void);
/* ABI version: bump on any signature change. */
int mzk_abi_version(void);

/* Size limits (tests/test_gpu_max_sizes.py runs them on one MI355X; profiles/r02u_max_sizes.txt has the timings):
 *   MSM / commit / open / SRS handles   n <= 2^27 pairs (MZK_E_ARG above; 2^27 pairs x 16 windows = 2^31 sort records, and an
 *                                       SRS handle with 16-bit window tables holds 16 n points = 128 GiB at 2^27 -- pass
 *                                       with_tables = 0 to mzk_srs_from_device_ex to keep only the prepared points)
 *   transforms                          n <= 2^28 over Fr (its 2-adicity); over M128 up to 2^32 nominally, memory-bound in
 *                                       practice: in + out + (passes - 1) twiddle tables of n elements each
 *   subproduct trees (mzk_fast_*)       n <= 2^27 points (the internal products are transforms of 2 n <= 2^28 points) */
/* ---- NTT family ------------------------------------------------------------------------------- */
/* ntt::ntt (algebra/ntt.rs:7-48) when inverse == 0: out[k] = sum_j in[j] * root^(j k), natural order
 * in and out, n a power of two, root a primitive n-th root.  n <= 1 copies the input.
 * ntt::intt (ntt.rs:50-64) when inverse != 0: n^-1 * ntt(root^-1, in); n == 1 copies the input.
 * `root` is the forward root in both cases, exactly as the reference's callers pass it. */
int mzk_ntt(int field_id, const uint64_t* root, const uint64_t* in, uint64_t* out, size_t n, int inverse);

/* ntt::fast_coset_evaluate (ntt.rs:254-269) incl. Polynomial::scale (polynomial.rs:167-174):
 * evaluations of the polynomial on { offset * generator^i : i < order }.  n_coef > order is the
 * reference's usize underflow -> MZK_E_LENGTH. */
int mzk_coset_lde(int field_id, const uint64_t* coef, size_t n_coef, const uint64_t* offset,
                  const uint64_t* generator, uint64_t* out, size_t order);

/* Polynomial::fft_multiply (polynomial.rs:242-276): omega must have order next_pow2(la+lb-1) (the
 * reference does not check it; neither do we).  out holds la+lb-1 elements; *out_len = length after
 * the reference's trailing-zero trim. */
int mzk_fft_multiply(int field_id, const uint64_t* a, size_t la, const uint64_t* b, size_t lb,
                     const uint64_t* omega, uint64_t* out, size_t* out_len);

/* ntt::fast_multiply (ntt.rs:66-116): root of order root_order; picks the sub-root itself; schoolbook
 * below degree 8; result NOT trimmed on the NTT path (length = chosen order).  out must hold
 * max(root_order, la+lb) elements. */
int mzk_fast_multiply(int field_id, const uint64_t* a, size_t la, const uint64_t* b, size_t lb,
                      const uint64_t* root, size_t root_order, uint64_t* out, size_t* out_len);

/* get_nth_root_of_m128 (zkstark/fri.rs:423-447) and the Fr analogue the reference lacks
 * (omega_2^28 = 5^((r-1)/2^28)).  Pure host parameter math. */
int mzk_root_of_unity(int field_id, unsigned log2_n, uint64_t* out);

/* ---- MSM / KZG -------------------------------------------------------------------------------- */
/* Polynomial::eval_with_powers_on_curve (algebra/polynomial.rs:156-165) = commit_kzg
 * (algebra/kzg.rs:57-59): sum_i scalars[i] * points[i].  Scalars need not be canonical (the
 * reference sanitizes, polynomial.rs:162): any 256-bit value is reduced mod r.  n == 0 -> infinity.
 * Points must lie on y^2 = x^3 + 3 (every G1Point the reference produces does): the kernels use the curve's
 * endomorphism (k P = k1 P + k2 phi(P), GLV), an identity of the prime-order group, not of arbitrary (x, y) pairs. */
int mzk_msm_g1_bn254(const uint64_t* scalars, const uint64_t* points_xy, size_t n, uint64_t out_xy[8]);

/* setup_kzg (kzg.rs:27-40), G1 part, trapdoor supplied by the caller: powers[i] = alpha^i * g1,
 * i = 0..=max_d  ((max_d+1) * 8 limbs out). */
int mzk_kzg_setup_g1(const uint64_t alpha[4], const uint64_t g1_xy[8], size_t max_d, uint64_t* powers_xy);

/* open_kzg (kzg.rs:61-72): y = f(u); w = MSM((f - y)/(X - u), powers). */
int mzk_kzg_open(const uint64_t* coef, size_t n, const uint64_t u[4], const uint64_t* powers_xy,
                 uint64_t y[4], uint64_t w_xy[8]);

/* ---- "next" rows of the path -------------------------------------------------------- */
/* batch_open_kzg (kzg.rs:74-88): ys[i] = f(us[i]); w = MSM((f - I)/prod(X - us[i]), powers) where I
 * interpolates (us, ys).  The us must be distinct (the reference's interpolate divides by their
 * differences).  ys: k*4 limbs out. */
int mzk_kzg_batch_open(const uint64_t* coef, size_t n, const uint64_t* us, size_t k, const uint64_t* powers_xy,
                       uint64_t* ys, uint64_t w_xy[8]);
/* prove_degree_bound (kzg.rs:121-134): MSM(f * X^(max_d - d), powers), max_d = n_powers - 1.
 * d > max_d (usize underflow) or a product longer than the SRS (index panic) -> MZK_E_LENGTH. */
int mzk_kzg_prove_degree_bound(const uint64_t* coef, size_t n, const uint64_t* powers_xy, size_t n_powers,
                               size_t d, uint64_t out_xy[8]);
/* ntt::fast_coset_divide (algebra/ntt.rs:271-330; FastStark::prove's quotients, zkstark/fast_stark.rs:265):
 * both polynomials scaled onto the coset offset*<root>, forward transforms, pointwise el * r.inverse()
 * (inverse(0) = 0, field.rs:209-232), inverse transform, first lhs.degree() - rhs.degree() + 1 coefficients scaled
 * back by offset^-1.  degree < 8 returns lhs / rhs (true long division, trimmed).  The reference's assertions map to
 * MZK_E_ROOT_ORDER / MZK_E_ROOT_PRIM (ntt.rs:282-283), MZK_E_ARG (rhs zero, :284), MZK_E_LENGTH (rhs.degree() >=
 * lhs.degree(), :285 -- a zero lhs included).  out must hold ll coefficients; *out_len receives the count. */
int mzk_fast_coset_divide(int field_id, const uint64_t* lhs, size_t ll, const uint64_t* rhs, size_t lr, const uint64_t* offset,
                          const uint64_t* root, size_t root_order, uint64_t* out, size_t* out_len);

/* ntt::fast_zerofier / fast_evaluate / fast_interpolate (algebra/ntt.rs:118-252; FastStark::prove interpolates every
 * trace register with fast_interpolate, zkstark/fast_stark.rs:209, and builds its transition zerofier with
 * fast_zerofier, :53).  Subproduct trees on the device (O(n log^2 n); the reference's remainders are O(n^2)).
 *   zerofier:    prod (X - domain[i]).  n = 0 -> empty; n < 8 -> n + 1 coefficients (the schoolbook branch trims);
 *                n >= 8 -> next_pow2(n + 1) coefficients, zero padded (fast_multiply does not trim its NTT branch).
 *                out must hold max(n + 1, next_pow2(n + 1)) elements.  Over the first n points of a power-of-two subgroup with
 *                at most 64 of its points missing (FastStark's transition zerofier, fast_stark.rs:53-57) the coefficients
 *                come from one launch: Z = (X^N - 1) / prod over the missing points, expanded by partial fractions.
 *   evaluate:    out[i] = f(domain[i]), i < n (m coefficients, any m).
 *   interpolate: the polynomial of degree < n through (domain[i], values[i]), trimmed like the reference's final sum;
 *                n = 1 -> [values[0]] untrimmed.  A repeated domain point gets weight inverse(0) = 0, as in the
 *                reference (field.rs:209-232 via ntt.rs:233-242).  out must hold n elements.
 *                When the domain is the first n points 1, g, g^2 ... of a subgroup of order next_pow2(n) with at most 64 of its
 *                points missing -- FastStark's trace_domain = omicron^i (fast_stark.rs:197-215) -- the same coefficients come from
 *                ONE inverse transform of the values extended by the interpolant's values at the missing points (a dot product
 *                each, weights kept with the cached plan); every other domain goes through the tree.
 * root / root_order: the two reference assertions (MZK_E_ROOT_ORDER / MZK_E_ROOT_PRIM); where the reference's
 * internal zerofier products would exceed root_order (it then wraps around or panics inside ntt) -> MZK_E_LENGTH. */
int mzk_fast_zerofier(int field_id, const uint64_t* domain, size_t n, const uint64_t* root, size_t root_order, uint64_t* out, size_t* out_len);
int mzk_fast_evaluate(int field_id, const uint64_t* coef, size_t m, const uint64_t* domain, size_t n, const uint64_t* root, size_t root_order,
                      uint64_t* out);
int mzk_fast_interpolate(int field_id, const uint64_t* domain, const uint64_t* values, size_t n, const uint64_t* root, size_t root_order,
                         uint64_t* out, size_t* out_len);
/* fast_interpolate of `batch` value vectors over ONE domain -- the registers of a trace (fast_stark.rs:203-215 calls it
 * once per register with the same trace_domain): the subproduct tree and the derivative values Z'(d_i), ~80 % of one
 * interpolation, are built once, and the registers share the launches of the up-sweep.  values: batch * n elements; out: batch rows of n elements (row r holds out_lens[r]
 * coefficients, zero-padded).  Each row bit-identical to the single call. */
int mzk_fast_interpolate_batch(int field_id, const uint64_t* domain, const uint64_t* values, size_t n, size_t batch, const uint64_t* root,
                               size_t root_order, uint64_t* out, size_t* out_lens);
/* The same with the values and the coefficients in HBM -- a prover uploads its trace once and the coefficients feed mzk_coset_lde_batch_dev
 * without visiting the host (fast_stark.rs:209-231: interpolate, then fast_coset_evaluate, per register).  d_values: batch rows of n
 * elements, canonical (not checked), complete on `stream` (a hipStream_t) before the call; d_out: batch rows of n elements, row r holds
 * out_lens[r] coefficients and zeros behind them.  domain and out_lens are host memory (the domain keys the cached plan, the lengths are
 * what the caller builds its Polynomial values from).  Returns when d_out is complete.  Rows bit-identical to mzk_fast_interpolate_batch. */
int mzk_fast_interpolate_batch_dev(int field_id, const uint64_t* domain, const void* d_values, size_t n, size_t batch, const uint64_t* root,
                                   size_t root_order, void* d_out, size_t* out_lens, void* stream);

/* FRI commit-loop split-and-fold (zkstark/fri.rs:182-193):
 * out[i] = 2^-1 ((1 + alpha/(offset omega^i)) c[i] + (1 - alpha/(offset omega^i)) c[n/2 + i]), i < n/2,
 * sanitized.  offset must be non-zero and omega a root of order n (as FRI::commit maintains). */
int mzk_fri_fold(int field_id, const uint64_t* codeword, size_t n, const uint64_t* alpha, const uint64_t* offset,
                 const uint64_t* omega, uint64_t* out);
int mzk_fri_fold_dev(int field_id, const void* d_codeword, size_t n, const uint64_t* alpha_host,
                     const uint64_t* offset_host, const uint64_t* omega_host, void* d_out, void* stream);

/* ---- SHA3-256 Merkle trees over codewords (algebra/merkle.rs:15-46 as called from zkstark/fri.rs:160-166,
 * 236-249 and fast_stark.rs:64-68, 237-241): leaves are bincode(FiniteFieldElement) of the canonical elements and
 * are not hashed themselves; a one-leaf tree commits to the leaf bytes.  n = 0 is an error (the reference recurses
 * forever).  Any n >= 1 is accepted like merkle.rs:15-25 does (mid = len / 2): the provers only commit to
 * power-of-two codewords, which take the fast path; other counts are reduced to the power-of-two tree over the
 * 2^floor(log2 n) one- or two-leaf subtrees at the bottom.  Merkle::open on a ragged tree only terminates when its
 * descent ends on a two-leaf slice (merkle.rs:32-34); for the one-leaf half of a three-leaf slice it recurses
 * forever, and mzk_merkle_open returns MZK_E_LENGTH there.  Paths have floor(log2 n) or floor(log2 n) + 1 entries.  A handle keeps its own copy of the leaves
 * and all node levels in HBM so that many paths can be opened against one codeword. */
typedef struct mzk_merkle mzk_merkle;
int mzk_merkle_build_field(int field_id, const uint64_t* elems, size_t n, mzk_merkle** out);
int mzk_merkle_build_field_dev(int field_id, const void* d_elems, size_t n, mzk_merkle** out, void* stream);
/* generic Merkle::commit input: leaf i = leaves[offsets[i] .. offsets[i+1]) (n + 1 offsets), any lengths */
int mzk_merkle_build_bytes(const uint8_t* leaves, const uint64_t* offsets, size_t n, mzk_merkle** out);
/* root: 32 bytes, or the leaf itself when n == 1 (merkle.rs:17-19); cap = capacity of `root` */
int mzk_merkle_root(const mzk_merkle* tree, uint8_t* root, size_t cap, size_t* root_len);
/* Merkle::open (merkle.rs:28-46): entry k of the bottom-up path goes to path + k * stride, its length to
 * path_len[k] (entry 0 is the sibling leaf verbatim, the rest are 32-byte digests); *depth = log2 n entries. */
int mzk_merkle_open(const mzk_merkle* tree, size_t index, uint8_t* path, size_t stride, uint64_t* path_len, size_t* depth);
/* `count` openings of one tree in one gather and one copy -- the FRI query phase opens three indices per colinearity test
 * and round (fri.rs:211-260; the reference's Merkle::open re-hashes the whole tree for each).  Path q, entry l:
 * paths[(q * depth + l) * stride ..], length path_lens[q * depth + l]; *depth entries per path, each path exactly what
 * mzk_merkle_open returns for indices[q].  Trees over field elements with a power-of-two leaf count (every codeword the
 * provers commit to); byte-leaf and ragged trees: MZK_E_ARG, open those one by one. */
int mzk_merkle_open_batch(const mzk_merkle* tree, const uint64_t* indices, size_t count, uint8_t* paths, size_t stride, uint64_t* path_lens,
                          size_t* depth);
/* Openings of SEVERAL trees in one call -- the query phase of FRI::prove (fri.rs:127-137 with reveal, :211-260) opens the a / b
 * indices in round i's tree and the c indices in round i+1's, for every round: with the trees of mzk_fri_commit_keep_trees one
 * call, one copy back and one synchronisation instead of two of each per round.  Tree t takes the next counts[t] entries of
 * `indices`; its paths follow tree t-1's in paths / path_lens, each tree laid out as mzk_merkle_open_batch lays it out, with
 * depths[t] entries per path.  counts[t] == 0 skips tree t.  Same kinds of trees as mzk_merkle_open_batch. */
int mzk_merkle_open_multi(const mzk_merkle* const* trees, size_t n_trees, const uint64_t* indices, const size_t* counts, uint8_t* paths,
                          size_t stride, uint64_t* path_lens, size_t* depths);
void mzk_merkle_free(mzk_merkle* tree);
/* one-shot Merkle::commit(codeword.map(bincode::serialize)) */
int mzk_merkle_commit_field(int field_id, const uint64_t* elems, size_t n, uint8_t* root, size_t cap, size_t* root_len);
int mzk_merkle_commit_field_dev(int field_id, const void* d_elems, size_t n, uint8_t* root_host, size_t cap, size_t* root_len,
                                void* stream);
/* Merkle::commit (merkle.rs:15-25) of `batch` codewords of n elements each, stored back to back -- the per-register loop of
 * the provers (fast_stark.rs:231-243: fast_coset_evaluate, serialize, Merkle::commit for every register) as one call.
 * roots: batch * 32 bytes (host).  n must be a power of two >= 2 (every codeword the provers commit to is; other leaf
 * counts: the single-tree calls).  The upper levels of a tree are one dependent hash per level on a nearly empty GPU; side
 * by side the trees share them: 16 codewords of 2^16 elements in 0.26 ms instead of 16 x 0.18. */
int mzk_merkle_commit_field_batch(int field_id, const uint64_t* elems, size_t n, size_t batch, uint8_t* roots);
int mzk_merkle_commit_field_batch_dev(int field_id, const void* d_elems, size_t n, size_t batch, uint8_t* roots, void* stream);
int mzk_merkle_commit_bytes(const uint8_t* leaves, const uint64_t* offsets, size_t n, uint8_t* root, size_t cap, size_t* root_len);

/* Elements whose BigInt the reference left NEGATIVE (F6: `%` keeps the sign, field.rs:98-110; only sanitize() folds
 * them into [0, p)): bincode writes Sign::Minus and the MAGNITUDE, so the leaf differs from the canonical element's.
 * The *_signed forms take the magnitudes (canonical limbs of |v|, |v| < p) plus one byte per element (1 = negative)
 * and hash exactly those bytes; all arithmetic afterwards uses the canonical representative p - |v|.
 * mzk_fri_commit_signed: round 0 commits to the unsanitized initial codeword like fri.rs:160-166 does (every later
 * codeword is sanitized by the fold, fri.rs:190); codewords_out[0..n) receives the canonical values. */
int mzk_merkle_build_field_signed(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, mzk_merkle** out);
int mzk_merkle_commit_field_signed(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, uint8_t* root, size_t cap,
                                   size_t* root_len);

/* FRI::commit (zkstark/fri.rs:144-209), codewords resident in HBM across all rounds.  Round r: the Merkle root of
 * codeword_r is handed to `challenge` (which owns the proof stream: push the root, and unless `last` sample
 * alpha = F::sample(prover_fiat_shamir(32)) into alpha_out, canonical limbs); then split-and-fold, omega and
 * offset squared.  roots: num_rounds x 48 bytes (root_len[r] = 32, or the leaf length once a codeword has one
 * element); codewords_out: the num_rounds codewords concatenated (n + n/2 + ... elements). */
/* The callback returns 0 on success; any other value aborts the loop with MZK_E_CALLBACK (a transcript error must
 * never let the prover fold with a default challenge).  alpha_out is pre-filled with all-ones (non-canonical). */
typedef int (*mzk_fri_challenge_fn)(void* user, int round, int last, const uint8_t* root, size_t root_len, uint64_t* alpha_out);
int mzk_fri_commit(int field_id, const uint64_t* codeword, size_t n, const uint64_t* omega, const uint64_t* offset, int num_rounds,
                   mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out);
int mzk_fri_commit_signed(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, const uint64_t* omega, const uint64_t* offset,
                          int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out);
/* The same loop (negative may be NULL: then `magnitudes` are the canonical elements), and the Merkle tree of every round
 * stays on the device for the query phase: trees_out[r] (num_rounds handles, NULL for a one-element round) opens with
 * mzk_merkle_open / mzk_merkle_open_batch exactly as Merkle::open on that round's codeword would (fri.rs:211-260); free each
 * with mzk_merkle_free. */
int mzk_fri_commit_keep_trees(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, const uint64_t* omega,
                              const uint64_t* offset, int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots,
                              uint64_t* root_len, uint64_t* codewords_out, mzk_merkle** trees_out);

/* mzk_fri_commit_keep_trees with the initial codeword already in HBM (the output of mzk_coset_lde_dev), complete before the call;
 * negative: host flags as in mzk_fri_commit_signed, or NULL.  In both keep-trees forms codewords_out may be NULL: the codewords
 * then never leave the device -- the query phase takes what FRI::reveal sends (fri.rs:211-260) from the trees: the paths with
 * mzk_merkle_open_multi, the a / b / c values and the last codeword with mzk_merkle_leaves. */
int mzk_fri_commit_keep_trees_dev(int field_id, const void* d_magnitudes, const uint8_t* negative, size_t n, const uint64_t* omega,
                                  const uint64_t* offset, int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots,
                                  uint64_t* root_len, uint64_t* codewords_out, mzk_merkle** trees_out);
/* The elements a field-element tree was built over, at `count` positions: magnitudes (count x limbs) and, if negative != NULL,
 * their Sign::Minus flags (all 0 unless the tree was built from signed leaves). */
int mzk_merkle_leaves(const mzk_merkle* tree, const uint64_t* indices, size_t count, uint64_t* magnitudes, uint8_t* negative);

/* ---- G2 (BN254 twist over Fq2 = Fq[u]/(u^2+1); bn128.rs:33-49) ------------------------------------------------
 * A G2 point is 16 limbs: x.c0 | x.c1 | y.c0 | y.c1 (4 limbs each, canonical); all-zero = infinity.
 * mzk_msm_g2_bn254: Polynomial::eval_with_powers_on_curve over pk.powers_2 (polynomial.rs:156-165 as called from
 * batch_verify_kzg, kzg.rs:114).  mzk_kzg_setup_g2: powers_2 of setup_kzg_with_full_g2 (kzg.rs:42-55) for a
 * caller-supplied alpha: [alpha^i] g2, i = 0..=max_d ((max_d + 1) x 16 limbs). */
int mzk_msm_g2_bn254(const uint64_t* scalars, const uint64_t* points_xy, size_t n, uint64_t out_xy[16]);
int mzk_msm_g2_bn254_dev(const void* d_scalars, const void* d_points_xy, size_t n, void* d_out_xy, void* stream);
int mzk_kzg_setup_g2(const uint64_t alpha[4], const uint64_t g2_xy[16], size_t max_d, uint64_t* powers2_xy);

/* Device-resident SRS for repeated commits against one PublicKeyKZG.powers_1 (kzg.rs:8-11). */
typedef struct mzk_srs mzk_srs;
int mzk_srs_upload(const uint64_t* powers_xy, size_t n, mzk_srs** out);
void mzk_srs_free(mzk_srs* srs);
int mzk_kzg_commit_srs(const mzk_srs* srs, const uint64_t* coef, size_t n, uint64_t out_xy[8]);
/* Persistence ("an SRS dump for PublicKeyKZG"; the reference only derives serde on its types and
 * never writes a key).  File = 64-byte header | n points exactly as at this ABI (x || y, little-endian u64 limbs,
 * canonical, all-zero = infinity) | optionally the window tables in the library's internal encoding; the header
 * holds an FNV-1a 64 of the point bytes, checked on load (MZK_E_IO).  mzk_srs_save(with_tables != 0) stores the tables
 * if the handle has them; mzk_srs_load(with_tables as in mzk_srs_from_device_ex) reads stored tables of the wanted
 * width and otherwise rebuilds them (which is faster than reading them from disk: see DESIGN.md).
 * mzk_srs_download returns powers_1 (n * 8 limbs) of a handle, e.g. one built by mzk_kzg_setup_g1_dev. */
int mzk_srs_save(const mzk_srs* srs, const char* path, int with_tables);
int mzk_srs_load(const char* path, int with_tables, mzk_srs** out);
int mzk_srs_download(const mzk_srs* srs, uint64_t* powers_xy, size_t cap_points);
size_t mzk_srs_len(const mzk_srs* srs);

/* ---- device-resident variants (inputs already in HBM; `stream` is a hipStream_t) --------------- */
int mzk_ntt_dev(int field_id, const uint64_t* root_host, const void* d_in, void* d_out, size_t n,
                int inverse, void* stream);
/* `batch` transforms of n points each, stored back to back (batch * n elements), all with the same root: what a prover
 * does column by column in a loop (ntt.rs:7-64 per column; fast_stark.rs interpolates and extends every register) as ONE
 * launch per pass, so that small transforms fill the GPU (2^12 points: 27 us for one transform, and about the same for
 * 64 of them).  Results are bit-identical to `batch` single calls.  In place allowed (d_in == d_out). */
int mzk_ntt_batch(int field_id, const uint64_t* root, const uint64_t* in, uint64_t* out, size_t n, size_t batch, int inverse);
int mzk_ntt_batch_dev(int field_id, const uint64_t* root, const void* d_in, void* d_out, size_t n, size_t batch, int inverse,
                      void* stream);
int mzk_coset_lde_dev(int field_id, const void* d_coef, size_t n_coef, const uint64_t* offset_host,
                      const uint64_t* generator_host, void* d_out, size_t order, void* stream);
/* `cols` transforms (ntt / intt, ntt.rs:7-64) of n_points = 2, 4, 8 or 16 points each, stored COLUMN-major: element i of
 * transform c at [i * cols + c], in and out (not in place).  One trip over the data; the step across the ranks of a transform
 * sharded over several GPUs (myzkp_amd/sharded.py), where after the first exchange a rank holds [source rank][its slice]. */
int mzk_ntt_columns_dev(int field_id, const uint64_t* root, const void* d_in, void* d_out, size_t n_points, size_t cols, int inverse,
                        void* stream);
/* Polynomial::scale (polynomial.rs:167-174) with an optional leading constant: out[i] = lead * coef[i] * ratio^i (lead == NULL: 1);
 * in place allowed.  Also the twiddle step between the local transforms of a transform sharded over several GPUs
 * (myzkp_amd/sharded.py: ratio = w^rank) and its n^-1 (ratio = 1). */
int mzk_poly_scale(int field_id, const uint64_t* coef, size_t n, const uint64_t* ratio, const uint64_t* lead, uint64_t* out);
int mzk_poly_scale_dev(int field_id, const void* d_coef, size_t n, const uint64_t* ratio_host, const uint64_t* lead_host, void* d_out,
                       void* stream);
/* fast_coset_evaluate (ntt.rs:254-269) of `batch` polynomials of n_coef coefficients each onto ONE coset -- the low-degree
 * extension of every column of a trace (fast_stark.rs:231,282,329 call it per polynomial) -- in one launch per pass:
 * coefs = batch * n_coef elements back to back, out = batch * order.  Bit-identical to `batch` single calls. */
int mzk_coset_lde_batch(int field_id, const uint64_t* coefs, size_t n_coef, const uint64_t* offset, const uint64_t* generator,
                        uint64_t* out, size_t order, size_t batch);
int mzk_coset_lde_batch_dev(int field_id, const void* d_coefs, size_t n_coef, const uint64_t* offset_host, const uint64_t* generator_host,
                            void* d_out, size_t order, size_t batch, void* stream);
/* d_out_xy: 8 limbs on the device */
int mzk_msm_g1_bn254_dev(const void* d_scalars, const void* d_points_xy, size_t n, void* d_out_xy, void* stream);
/* Multi-GPU sharding (one process per GPU): each rank reduces its shard to one XYZZ partial
 * (16 limbs, internal Montgomery encoding, opaque), the partials are exchanged with an all-gather
 * (RCCL has no elliptic-curve reduction operator), and every rank folds them. */
int mzk_msm_g1_bn254_partial_dev(const void* d_scalars, const void* d_points_xy, size_t n,
                                 void* d_partial16, void* stream);
int mzk_g1_fold_partials_dev(const void* d_partials16, int count, void* d_out_xy, void* stream);
/* ---- the same sharding inside ONE process (contexts of mzk_init_devices; BASELINE configs[3] without Python) ----
 * Context r owns the contiguous slice mzk_shard_range(n, r, world) of scalars and points, reduces it to one XYZZ
 * partial on its own GPU; the W 128-byte records are gathered on context 0 (pinned host buffer: no peer access or
 * collective library needed for 128 bytes per GPU) and folded there.  All W pipelines run concurrently.  Blocking. */
void mzk_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi);
int mzk_msm_g1_bn254_multi(const uint64_t* scalars, const uint64_t* points_xy, size_t n, uint64_t out_xy[8]);
/* ONE n-point transform (ntt / intt, ntt.rs:7-64; natural order) whose vector is spread over the W contexts -- the four-step
 * layout of SURVEY 8e: W-point transforms across the GPUs (mzk_ntt_columns_dev), the local n/W-point transform with the
 * twiddle fused into it (fast_coset_evaluate with offset root^rank), and all-to-all exchanges between them (peer copies,
 * each destination pulling its chunk of every source).  Layouts of a distributed vector: CONTIGUOUS = part r is
 * x[r n/W, (r+1) n/W); CYCLIC = part r is x[r], x[r + W], ...  contiguous -> cyclic and cyclic -> contiguous cost two
 * exchanges, contiguous -> contiguous three (cyclic -> cyclic: MZK_E_ARG); keep the cyclic layout between a forward and an
 * inverse transform.  W a power of two <= 16 with W^2 <= n.  d_in_parts[r] / d_out_parts[r]: n/W elements on context r's
 * GPU, inputs complete before the call, not overwritten; blocking.  One process per GPU: myzkp_amd/sharded.py runs the
 * same schedule over RCCL's all-to-all. */
enum { MZK_LAYOUT_CONTIGUOUS = 0, MZK_LAYOUT_CYCLIC = 1 };
int mzk_ntt_multi_dev(int field_id, const uint64_t* root, const void* const* d_in_parts, void* const* d_out_parts, size_t n, int inverse,
                      int layout_in, int layout_out);
int mzk_ntt_multi(int field_id, const uint64_t* root, const uint64_t* in, uint64_t* out, size_t n, int inverse);
typedef struct mzk_srs_multi mzk_srs_multi;
/* PublicKeyKZG.powers_1 sharded over the contexts: from host points, or built on the GPUs (setup_kzg, kzg.rs:27-40:
 * context r computes powers [lo_r, hi_r) itself; with_tables as in mzk_srs_from_device_ex). */
int mzk_srs_upload_multi(const uint64_t* powers_xy, size_t n, mzk_srs_multi** out);
int mzk_kzg_setup_srs_multi(const uint64_t alpha[4], const uint64_t g1_xy[8], size_t max_d, int with_tables, mzk_srs_multi** out);
void mzk_srs_multi_free(mzk_srs_multi* h);
int mzk_srs_multi_world(const mzk_srs_multi* h);
size_t mzk_srs_multi_shard_lo(const mzk_srs_multi* h, int rank);   /* rank = world gives n */
/* commit_kzg (kzg.rs:57-59) over the sharded SRS: coefficients in host memory, or d_coef_shards[r] = device pointer
 * on context r's GPU to coefficients [lo_r, min(hi_r, n)), complete before the call. */
int mzk_kzg_commit_srs_multi(const mzk_srs_multi* h, const uint64_t* coef, size_t n, uint64_t out_xy[8]);
int mzk_kzg_commit_srs_multi_dev(const mzk_srs_multi* h, const void* const* d_coef_shards, size_t n, uint64_t out_xy[8]);

/* commit against a device-resident SRS with device-resident coefficients; out_partial != 0 writes the
 * 16-limb XYZZ partial (multi-GPU shard) instead of the 8-limb affine point. */
int mzk_kzg_commit_srs_dev(const mzk_srs* srs, const void* d_coef, size_t n, void* d_out, int out_partial,
                           void* stream);

/* `count` commitments against one SRS -- commit_kzg (kzg.rs:57-59) once per polynomial, as the reference's callers do in
 * a loop -- with results bit-identical to `count` single calls.  coefs: count vectors of n coefficients back to back
 * (count * n * 4 limbs), out_xy: count affine points (count * 8 limbs).  One commit is kept in flight per context of the
 * current GPU (at most four; mzk_init_devices with the same ordinal repeated makes them: own stream and workspace each,
 * the SRS handle shared), so the latency-bound tail of one commit runs under the bucket accumulation of the others:
 * 1.40 instead of 1.64 ms per 2^20-coefficient commit with four contexts.  With a single context the commits simply
 * run one after the other.  The _dev form forks from and joins `stream`; max_in_flight caps the contexts it uses
 * (0 = up to four: more measured slower except for the smallest commitments; an explicit value may go up to eight).  The
 * lanes want hardware queues of their own: with more than four streams alive on the GPU set GPU_MAX_HW_QUEUES=8. */
int mzk_kzg_commit_srs_batch(const mzk_srs* srs, const uint64_t* coefs, size_t n, size_t count, uint64_t* out_xy);
int mzk_kzg_commit_srs_batch_dev(const mzk_srs* srs, const void* d_coefs, size_t n, size_t count, void* d_out_xy, int max_in_flight,
                                 void* stream);
/* open_kzg (kzg.rs:61-72) of `count` polynomials of n coefficients each against one SRS, polynomial i at us[4 i .. 4 i + 4)
 * (repeat the point to open all at one place): d_ys = count * 4 limbs, d_ws_xy = count * 8 limbs, both on the device.  Same
 * lanes as the batch commit; every (y_i, w_i) bit-identical to mzk_kzg_open_srs_dev. */
int mzk_kzg_open_srs_batch_dev(const mzk_srs* srs, const void* d_coefs, size_t n, size_t count, const uint64_t* us, void* d_ys, void* d_ws_xy,
                               int max_in_flight, void* stream);
/* the same with coefficients and results in host memory (blocking): ys = count * 4 limbs, ws_xy = count * 8 limbs */
int mzk_kzg_open_srs_batch(const mzk_srs* srs, const uint64_t* coefs, size_t n, size_t count, const uint64_t* us, uint64_t* ys, uint64_t* ws_xy);
/* The reference's actual call pattern -- hundreds of SHORT polynomials against one `pk`: one commit_kzg per row
 * (das/avail.rs:88-98), per chunk (das/eigenda.rs:92-101), per folded polynomial (algebra/gemini.rs:112-114), one open_kzg
 * per cell (das/avail.rs:132).  When the handle holds narrow window tables (8- or 10..13-bit: the default up to 2^14 points)
 * and count > 1, the whole batch runs as ONE bucket problem over (polynomial x bucket) -- digit sort, accumulation, bucket
 * reduction and one tail workgroup PER POLYNOMIAL, the same handful of launches whatever `count` is -- instead of one commit
 * per lane: 256 commitments of 2^10 coefficients take about what six single calls took (0.76 ms; 0.52 - 0.59 ms over the
 * direct tables below).  Arguments and results exactly as the _batch_dev forms, which route here themselves from four
 * polynomials on; handles without narrow tables take the lanes of the _batch forms.  Every point bit-identical to the single
 * call.  Work enqueued on `stream` only: no fork, no extra contexts needed. */
int mzk_kzg_commit_srs_many_dev(const mzk_srs* srs, const void* d_coefs, size_t n, size_t count, void* d_out_xy, void* stream);
/* Optional second table set of a SHORT SRS (at most 2^14 points) for those batches: every multiple a signed window digit can
 * ask for, D[w][i][m] = (m + 1) 2^(c w) P_i -- (254 / c + 1) n 2^(c-1) affine points: 0.85 GiB for 1024 powers at c = 10.  With
 * it a batch needs no buckets at all: one gathered row and one mixed addition per (coefficient, window), a tree per 256
 * coefficients and one fold per polynomial -- two launches, no digit sort, no bucket reduction (the large HBM is what makes
 * this affordable; build once per SRS like the window tables).  window_bits: 8..12, or 0 = the widest that fits max_bytes
 * (0 = a quarter of the free device memory, at most 4 GiB).  Blocking (the tables are complete on return); calling it again with
 * another width replaces them; mzk_srs_drop_direct releases them (mzk_srs_free does too).  Results are bit-identical with and
 * without: the same group element, converted to the same canonical affine point. */
int mzk_srs_build_direct(mzk_srs* srs, int window_bits, size_t max_bytes, void* stream);
void mzk_srs_drop_direct(mzk_srs* srs);
/* What a handle holds: window width of its tables (0 = none: prepared points only), of its direct tables (0 = none), and the
 * device bytes of both together. */
int mzk_srs_window_bits(const mzk_srs* srs);
int mzk_srs_direct_bits(const mzk_srs* srs);
/* Window width (bits of a signed digit) mzk_msm_g1_bn254* uses for n arbitrary points: every scalar is split in two halves of
 * 127 bits (mzk_glv.h), each cut into 126 / bits + 1 windows -- 2 x that many mixed additions per pair (16 bits and 16 additions
 * below 3 x 2^21 pairs, 19 bits and 14 additions from there on).  Sizing information for the callers of Polynomial::
 * eval_with_powers_on_curve (polynomial.rs:156-165), which has no such notion; bench.py prices the accumulate kernel with it. */
int mzk_msm_generic_window_bits(size_t n);
size_t mzk_srs_table_bytes(const mzk_srs* srs);
int mzk_kzg_open_srs_many_dev(const mzk_srs* srs, const void* d_coefs, size_t n, size_t count, const uint64_t* us, void* d_ys, void* d_ws_xy,
                              void* stream);

/* open_kzg / setup_kzg with everything device-resident (end-to-end pipelines: iNTT -> commit -> open).
 * d_y: 4 limbs, d_w_xy: 8 limbs on the device; u / alpha / g1 are host parameters. */
int mzk_kzg_open_srs_dev(const mzk_srs* srs, const void* d_coef, size_t n, const uint64_t u_host[4], void* d_y,
                         void* d_w_xy, void* stream);
int mzk_kzg_setup_g1_dev(const uint64_t alpha_host[4], const uint64_t g1_xy_host[8], size_t max_d, void* d_powers_xy,
                         void* stream);
/* Sharded variants (one process per GPU): rank g builds powers [first, first+count) of the SRS and MSMs its own
 * slice of the coefficient / quotient vectors; mzk_kzg_open_quotient_dev exposes y = f(u) (4 limbs) and the
 * quotient (f - y)/(X - u) (n-1 elements) so that every rank can commit its slice of it. */
int mzk_kzg_setup_g1_range_dev(const uint64_t alpha_host[4], const uint64_t g1_xy_host[8], size_t first, size_t count,
                               void* d_powers_xy, void* stream);
int mzk_kzg_open_quotient_dev(const void* d_coef, size_t n, const uint64_t u_host[4], void* d_y, void* d_q, void* stream);
/* The quotient of open_kzg (kzg.rs:61-72: (f - y) / (X - u)) SHARDED over the ranks: q_{i-1} = b_i with b_i = c_i + u b_{i+1} is a suffix
 * recurrence over the whole coefficient vector, so the rank that holds the slice c[lo, hi) needs one value from the ranks above it,
 * b_hi.  Two local passes around one exchange of 32 bytes per rank:
 *   mzk_kzg_open_slice_value_dev     d_value (4 limbs) = sum_t c[lo + t] u^t, the slice's value at u (its b_lo with carry 0);
 *   (the ranks gather the values; top rank down: b_lo(g) = value(g) + u^(hi - lo) b_hi(g), b_hi(g - 1) = b_lo(g); y = b_lo(0))
 *   mzk_kzg_open_slice_quotient_dev  d_q_slice (len elements) = b[lo + 1 .. hi], given carry_in = b_hi (4 limbs, canonical; 0 for the
 *                                    top rank) -- exactly the quotient coefficients q[lo .. hi) the rank commits against ITS powers
 *                                    (the top rank's last one, q[n - 1] = b_n = 0, contributes nothing).
 * Same field elements as the one-piece recurrence, hence the same witness.  myzkp_amd/sharded.py: sharded_open_quotient. */
int mzk_kzg_open_slice_value_dev(const void* d_coef_slice, size_t len, const uint64_t u_host[4], void* d_value, void* stream);
int mzk_kzg_open_slice_quotient_dev(const void* d_coef_slice, size_t len, const uint64_t u_host[4], const uint64_t carry_in[4], void* d_q_slice,
                                    void* stream);
/* Build an SRS handle from points already in HBM (affine canonical, n * 8 limbs).  The _ex form chooses
 * whether the window tables are built (worth it from ~30 commits per SRS on at 2^20 points; a one-shot pipeline keeps
 * the plain prepared points and pays the window Horner instead).  Default widths by size: 8 bits up to 1024 points,
 * 10 up to 2^14, 16 below 2^19, 17 below 2^22, 20 from there on (254 / c + 1 tables of n points each: 15 at 17 bits, 13 at 20;
 * measured per size, profiles/r03b_window_sweep.txt, profiles/round4_window_sweep.txt). */
int mzk_srs_from_device(const void* d_powers_xy, size_t n, mzk_srs** out, void* stream);
int mzk_srs_from_device_ex(const void* d_powers_xy, size_t n, int with_tables, mzk_srs** out, void* stream);
/* with_tables: 0 = plain prepared points, 1 = tables with the default window width for n (above), 8..22 = that window width
 * (tuning / tests, and BASELINE configs[2]'s 16 bits at any size).
 * A handle never fails for lack of TABLE memory (commit_kzg(&poly, &pk) does not either): when the wanted tables exceed the
 * budget set by mzk_set_table_budget (bytes per handle; 0 = no limit, the default) or the device refuses the allocation, the
 * handle degrades -- every 2nd table with two bucket sets, every 4th with four (from 2^15 points on; 1/2 and 1/4 of the
 * memory, the same additions, 2 - 4 times the bucket reduction and a few dozen doublings at the end), finally the prepared
 * points alone (the generic layout: 128 bytes per point, ~25 % slower commits) -- and every commitment stays bit-identical.
 * mzk_srs_window_bits / mzk_srs_bucket_sets / mzk_srs_table_bytes tell what a handle got (bucket sets: 1 = full tables,
 * 2 / 4 = degraded, 0 = no tables).  Applies to mzk_srs_upload, mzk_srs_from_device[_ex], mzk_srs_load and the sharded forms. */
int mzk_set_table_budget(size_t bytes);
int mzk_srs_bucket_sets(const mzk_srs* srs);
/* Scratch memory that gives itself back (the reference's Vec buffers are freed when commit_kzg / ntt return, kzg.rs:57-59,
 * ntt.rs:7-48; the library instead keeps per-context workspace buffers so that steady-state calls never allocate).
 *   mzk_set_workspace_budget(bytes): the workspace a context may KEEP (0 = no limit, the default).  A call takes what it needs;
 *     whenever a buffer has to grow while the context holds more than the budget, every buffer the running call has not asked for
 *     is released first (the device is waited for).  Setting a budget below what is held releases the idle buffers at once.
 *   Whatever the budget: an allocation the device refuses -- workspace, SRS tables, prepared points -- is tried again after the idle
 *     workspace has been released, and a new SRS handle does that BEFORE it degrades its table layout; what is still refused is
 *     MZK_E_NOMEM (never a bare MZK_E_HIP), with nothing enqueued.
 *   mzk_trim_workspace(&released): releases every workspace buffer of every context and the tables of the cached transform plans
 *     (rebuilt on demand); waits for the devices.  released may be NULL.
 *   mzk_workspace_bytes(&bytes): workspace bytes currently held over all contexts. */
int mzk_set_workspace_budget(size_t bytes);
int mzk_trim_workspace(size_t* bytes_released);
int mzk_workspace_bytes(size_t* bytes);
/* Device copy at 16 bytes per lane, grid-stride, on `stream` (d_dst and d_src 16-byte aligned, bytes a multiple of 16): the
 * memory-bound yardstick the benchmark quotes beside the nominal HBM peak (bench.py hbm_copy_GBps_measured). */
int mzk_selftest_copy_dev(const void* d_src, void* d_dst, size_t bytes, void* stream);

/* Host-side parameter arithmetic of the library (roots, inverses, offsets: O(log n) scalar work per call, never on the
 * data path), exposed for checking without a GPU: op 0 = a * b, 1 = a^-1 (0 -> 0), 2 = a^(b[0]), 3 = a * b by the
 * bit-serial reference implementation.  field_id may also be MZK_FIELD_FQ. */
int mzk_host_field_op(int field_id, int op, const uint64_t* a, const uint64_t* b, uint64_t* out);

/* Device self-check: the throughput kernels compute their Montgomery products with hand-scheduled inline-asm blocks
 * (myzkp_amd/csrc/mzk_field_asm.h); this runs both forms (asm and portable C++) over n operand sets per field -- random,
 * all-ones, zero and the widest lazy limbs -- and returns the number of differing results (must be 0).
 * field_id may also be MZK_FIELD_FQ. */
int mzk_selftest_field_asm(int field_id, uint64_t seed, size_t n, uint64_t* mismatches);
/* Device self-check of the ROW-cooperative group operations behind the MSM tails (myzkp_amd/csrc/mzk_row.h: one point
 * operation per wave, field elements spread over DPP rows): n pairs of XYZZ points -- independent points, P + P, P + (-P),
 * infinity on either side -- added and doubled `dbl_reps` times by the row code and by the plain exception-complete formulas
 * (the group law of curve.rs:44-161), compared as group elements; every record is also checked against the storage bound.
 * Returns the number of pairs that differ (must be 0). */
int mzk_selftest_row_ec(uint64_t seed, size_t n, int dbl_reps, uint64_t* mismatches);
/* Device self-check of the wave-cooperative Fq inversion that ends every MSM (myzkp_amd/csrc/mzk_inv_wave.h: the safegcd with
 * the limbs of f, g, d, e spread over the lanes): n values -- 0, 1, 2, 3, 2^29 - 1, 2^28, -1, 1 in Montgomery form, then random
 * ones -- against the single-lane safegcd (which the host build pins on the oracle) and against a * a^-1 == 1.  Returns the
 * number of values that differ (must be 0). */
int mzk_selftest_inv_wave(uint64_t seed, size_t n, uint64_t* mismatches);

/* Deterministic synthetic inputs (bench + tests): bit-identical to the oracle's orc_synth_*. */
int mzk_synth_field_dev(int field_id, uint64_t seed, size_t n, void* d_out, void* stream);
int mzk_synth_g1_points_dev(uint64_t seed, size_t n, void* d_out_xy, void* stream);

/* ---- per-phase device timing (HIP events recorded on the launch stream around each kernel group) ---- */
enum { MZK_PH_MSM_PREPARE = 0, MZK_PH_MSM_SORT = 1, MZK_PH_MSM_ACCUMULATE = 2, MZK_PH_MSM_REDUCE = 3,
       MZK_PH_MSM_COMBINE = 4, MZK_PH_NTT_PASS0 = 5, MZK_PH_NTT_PASS1 = 6, MZK_PH_NTT_PASS2 = 7,
       MZK_PH_NTT_PASS3 = 8, MZK_PH_NTT_PRESCALE = 9, MZK_PH_MERKLE = 10,
       MZK_PH_MSM_SEG_COMBINE = 11,   /* k_seg_combine (+ heavy): sums a bucket's segment partials; MSM_ACCUMULATE is k_seg_accumulate alone */
       MZK_PH_NTT_TOTAL = 12,         /* one pair around all passes of a transform */
       MZK_PH_COUNT = 13 };
int mzk_prof_enable(int on);
/* bit p set = phase p gets its event pair while profiling is on (default: all).  An event pair costs a few
 * microseconds of stream time, so a timed region instruments only the kernel it prices. */
int mzk_prof_select(uint32_t phase_mask);
int mzk_prof_reset(void);
/* Synchronises the device, folds all pending event pairs, returns accumulated ms and launch count. */
int mzk_prof_read(int phase, double* total_ms, uint64_t* launches);
const char* mzk_prof_name(int phase);

#ifdef __cplusplus
}
#endif
#endif /* MZK_H */
