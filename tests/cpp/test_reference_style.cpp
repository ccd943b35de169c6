// GPU tests that read like the reference's own #[test]s, written against the C++ mirror of the Rust
// surface (myzkp_amd/host/myzkp.hpp) and checked with the oracle (test infrastructure).
//   test_ntt            <- algebra/ntt.rs:346-374
//   test_kzg            <- algebra/kzg.rs:152-172 (pairing check replaced by the trapdoor identity)
//   test_g1             <- algebra/curve/bn128.rs:285-301 (relations, through the MSM)
//   test_fast_multiply  <- algebra/ntt.rs:66-116 vs Polynomial::fft_multiply
//   test_panics         <- ntt.rs:8-23, polynomial.rs:162 index panic
//   test_merkle         <- algebra/merkle.rs:76-93
//   test_g2             <- algebra/curve/bn128.rs:306-323 and algebra/kzg.rs:110-114
//   test_fri_commit     <- zkstark/fri.rs:144-209 (commit phase; transcript stood in for by a deterministic challenge)
//   test_batch_extensions  the batch:: forms against the loops of single calls they replace
#include <cstdio>
#include <cstdlib>
#include "../../myzkp_amd/host/myzkp.hpp"
using namespace myzkp;

extern "C" {  // oracle (checker only)
int orc_poly_eval(int fid, const uint64_t* coef, size_t n, const uint64_t* x, uint64_t* out);
int orc_field_pow(int fid, const uint64_t* a, const uint64_t* e, int ne, uint64_t* out);
int orc_field_mul(int fid, const uint64_t* a, const uint64_t* b, uint64_t* out);
int orc_ec_mul(int cid, const uint64_t* p_xy, const uint64_t* k, int nk, uint64_t* out_xy);
int orc_g2_mul(const uint64_t* p, const uint64_t* k, int nk, uint64_t* out);
int orc_merkle_verify_ref(const uint8_t* root, size_t root_len, size_t index, const uint8_t* path, const uint64_t* path_len, size_t stride,
                          size_t depth, const uint8_t* leaf, size_t leaf_len);
int orc_merkle_commit_ref(const uint8_t* leaves, const uint64_t* offsets, size_t n, uint8_t* root, size_t* root_len);
void orc_bincode_field_vector(const uint64_t* elems, int nl, size_t n, uint8_t* leaves, uint64_t* offsets);
int orc_fri_fold_ref(int fid, const uint64_t* codeword, size_t n, const uint64_t* alpha, const uint64_t* offset, const uint64_t* omega,
                     uint64_t* out);
void orc_synth_vector(int fid, uint64_t seed, size_t n, uint64_t* out, int nthreads);
}
static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

static void test_ntt() {
  const unsigned logn = 8;
  const size_t n = 1u << logn;
  auto primitive_root = get_nth_root_of_m128(logn);
  std::vector<FiniteFieldElement<M128>> coef;
  for (size_t i = 0; i < n; i++) coef.push_back(FiniteFieldElement<M128>::from_value(i + 1));
  Polynomial<FiniteFieldElement<M128>> poly{coef};
  auto values = ntt(primitive_root, coef);
  CHECK(values.size() == n);
  auto cw = to_wire(poly.coef);
  for (size_t i = 0; i < n; i++) {  // values_again = poly.eval_domain(root^i)
    uint64_t e = i, x[2], y[2];
    orc_field_pow(MZK_FIELD_M128, primitive_root.value.data(), &e, 1, x);
    orc_poly_eval(MZK_FIELD_M128, cw.data(), n, x, y);
    CHECK(values[i] == FiniteFieldElement<M128>::from_limbs(y));
  }
  auto coef_again = intt(primitive_root, values);
  CHECK(coef_again.size() == coef.size());
  for (size_t i = 0; i < n; i++) CHECK(coef[i] == coef_again[i]);
}

static void test_kzg() {
  auto g1 = BN128::generator_g1();
  // (x+1)(x+2)(x+3) = 6 + 11x + 6x^2 + x^3, kzg.rs:157-161
  Polynomial<FqOrder> f{{FqOrder::from_value(6), FqOrder::from_value(11), FqOrder::from_value(6), FqOrder::from_value(1)}};
  auto alpha = FqOrder::from_value(7);
  auto pk = setup_kzg_with_alpha(g1, alpha, 3);
  CHECK(pk.powers_1.size() == 4);
  auto c = commit_kzg(f, pk);
  // [f(7)] G, f(7) = 8*9*10 = 720
  uint64_t g[8], k[4] = {720, 0, 0, 0}, want[8];
  g1.to_wire(g);
  orc_ec_mul(0, g, k, 4, want);
  CHECK(c == G1Point::from_wire(want));
  auto u = FqOrder::from_value(5);
  auto proof = open_kzg(f, u, pk);
  CHECK(proof.y == FqOrder::from_value(6 * 7 * 8));
  // q(X) = (f(X) - f(5)) / (X - 5); q(7) = (720 - 336) / 2 = 192
  uint64_t kq[4] = {192, 0, 0, 0};
  orc_ec_mul(0, g, kq, 4, want);
  CHECK(proof.w == G1Point::from_wire(want));
  // tampered polynomial does not open to the same witness (kzg.rs:202-204 analogue)
  Polynomial<FqOrder> f2 = f;
  f2.coef[0] = FqOrder::from_value(7);
  CHECK(!(commit_kzg(f2, pk) == c));
}

static void test_batch_kzg() {   // kzg.rs:174-204 test_batch_kzg, pairing check replaced by the quotient identity
  auto g1 = BN128::generator_g1();
  Polynomial<FqOrder> f{{FqOrder::from_value(6), FqOrder::from_value(11), FqOrder::from_value(6), FqOrder::from_value(1)}};
  auto pk = setup_kzg_with_alpha(g1, FqOrder::from_value(7), 3);
  std::vector<FqOrder> us{FqOrder::from_value(5), FqOrder::from_value(9)};
  auto proof = batch_open_kzg(f, us, pk);
  CHECK(proof.ys.size() == 2);
  CHECK(proof.ys[0] == FqOrder::from_value(6 * 7 * 8));
  CHECK(proof.ys[1] == FqOrder::from_value(10 * 11 * 12));
  // f = (X-5)(X-9)(X+20) + I(X)  =>  quotient q(X) = X + 20, q(7) = 27
  uint64_t g[8], k[4] = {27, 0, 0, 0}, want[8];
  g1.to_wire(g);
  orc_ec_mul(0, g, k, 4, want);
  CHECK(proof.w == G1Point::from_wire(want));
  // degree bound: f has degree 3 <= 3 -> MSM(f * X^0); bound 2 does not fit the SRS
  CHECK(prove_degree_bound(f, pk, 3) == commit_kzg(f, pk));
  bool caught = false;
  try { prove_degree_bound(f, pk, 2); } catch (const Panic& p) { caught = p.code == MZK_E_LENGTH; }
  CHECK(caught);
}

static void test_g1() {
  auto g = BN128::generator_g1();
  auto mul = [&](uint64_t k) { return Polynomial<FqOrder>{{FqOrder::from_value(k)}}.eval_with_powers_on_curve({g}); };
  auto g2 = mul(2);
  // 2G + G + G == (2G) * 2
  auto lhs = Polynomial<FqOrder>{{FqOrder::one(), FqOrder::one(), FqOrder::one()}}.eval_with_powers_on_curve({g2, g, g});
  auto rhs = Polynomial<FqOrder>{{FqOrder::from_value(2)}}.eval_with_powers_on_curve({g2});
  CHECK(lhs == rhs);
  // 9G + 5G == 12G + 2G
  auto a = Polynomial<FqOrder>{{FqOrder::from_value(9), FqOrder::from_value(5)}}.eval_with_powers_on_curve({g, g});
  auto b = Polynomial<FqOrder>{{FqOrder::from_value(12), FqOrder::from_value(2)}}.eval_with_powers_on_curve({g, g});
  CHECK(a == b);
  // r * G == infinity, as (r-1) G + G
  uint64_t rm1[4] = {0x43e1f593f0000000ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  auto z = Polynomial<FqOrder>{{FqOrder::from_limbs(rm1), FqOrder::one()}}.eval_with_powers_on_curve({g, g});
  CHECK(z.is_point_at_infinity());
  CHECK(Polynomial<FqOrder>{}.eval_with_powers_on_curve({}).is_point_at_infinity());
}

static void test_fast_multiply() {
  Polynomial<FqOrder> a, b;
  for (uint64_t i = 0; i < 40; i++) a.coef.push_back(FqOrder::from_value(3 * i + 1));
  for (uint64_t i = 0; i < 25; i++) b.coef.push_back(FqOrder::from_value(i * i + 2));
  auto p1 = a.fft_multiply(b, get_nth_root_of_fr(6));   // 64 >= 40 + 25 - 1
  auto p2 = fast_multiply(a, b, get_nth_root_of_fr(10), 1024);
  CHECK(p1.coef.size() == 64);
  CHECK(p2.coef.size() == 64);   // untrimmed order (ntt.rs:90-93,113-115): degree 63 -> order 64
  for (size_t i = 0; i < p1.coef.size(); i++) CHECK(p1.coef[i] == p2.coef[i]);
  for (size_t i = p1.coef.size(); i < p2.coef.size(); i++) CHECK(p2.coef[i].is_zero());
  // spot check coefficient 1: a0 b1 + a1 b0 = 1*3 + 4*2
  CHECK(p1.coef[1] == FqOrder::from_value(11));
}

static void test_panics() {
  std::vector<FiniteFieldElement<M128>> v(6, FiniteFieldElement<M128>::one());
  bool caught = false;
  try { ntt(get_nth_root_of_m128(3), v); } catch (const Panic& p) {
    caught = std::string(p.what()) == "cannot compute ntt of non-power-of-two sequence";
  }
  CHECK(caught);
  v.resize(8, FiniteFieldElement<M128>::one());
  caught = false;
  try { ntt(get_nth_root_of_m128(4), v); } catch (const Panic& p) {
    caught = std::string(p.what()) == "primitive root must be nth root of unity, where n is len(values)";
  }
  CHECK(caught);
  caught = false;
  try { ntt(get_nth_root_of_m128(2), v); } catch (const Panic& p) {
    caught = std::string(p.what()) == "primitive root is not primitive nth root of unity, where n is len(values)";
  }
  CHECK(caught);
  caught = false;
  try {
    Polynomial<FqOrder>{{FqOrder::one(), FqOrder::one()}}.eval_with_powers_on_curve({BN128::generator_g1()});
  } catch (const Panic& p) { caught = p.code == MZK_E_LENGTH; }
  CHECK(caught);
}

static bool oracle_verify(const MerkleRoot& root, size_t index, const MerklePath& path, const std::vector<uint8_t>& leaf) {
  size_t stride = 32;
  for (auto& e : path) if (e.size() > stride) stride = e.size();
  std::vector<uint8_t> buf(stride * path.size());
  std::vector<uint64_t> lens;
  for (size_t k = 0; k < path.size(); k++) { std::copy(path[k].begin(), path[k].end(), buf.begin() + k * stride); lens.push_back(path[k].size()); }
  return orc_merkle_verify_ref(root.data(), root.size(), index, buf.data(), lens.data(), stride, path.size(), leaf.data(), leaf.size()) != 0;
}
static std::vector<uint8_t> bytes_of(const char* s) { return std::vector<uint8_t>(s, s + strlen(s)); }

static void test_merkle() {   // merkle.rs:76-93
  std::vector<std::vector<uint8_t>> leafs = {bytes_of("leaf1"), bytes_of("leaf2"), bytes_of("leaf3"), bytes_of("leaf4")};
  MerkleRoot root = Merkle::commit(leafs);
  size_t index = 2;
  MerklePath proof = Merkle::open(index, leafs);
  CHECK(oracle_verify(root, index, proof, leafs[index]));
  CHECK(!oracle_verify(root, index, proof, leafs[index + 1]));
}

static void test_fri_commit() {   // fri.rs:144-209
  typedef FiniteFieldElement<M128> F;
  const size_t n = 1 << 10;
  const int rounds = 5;
  std::vector<uint64_t> raw(2 * n);
  orc_synth_vector(MZK_FIELD_M128, 4242, n, raw.data(), 1);
  std::vector<F> cw;
  for (size_t i = 0; i < n; i++) cw.push_back(F::from_limbs(&raw[2 * i]));
  F omega = get_nth_root_of_m128(10);
  F offset = F::from_value(3);
  std::vector<MerkleRoot> seen;
  auto challenge = [&](int round, bool last, const MerkleRoot& root) {
    seen.push_back(root);
    F a = F::from_value(0);
    if (!last) { a.value[0] = 0x1234567ULL * (round + 1) + root[0]; a.value[1] = root[1]; }   // < 2^72: canonical
    return a;
  };
  auto res = fri_commit(cw, omega, offset, rounds, challenge);
  CHECK(res.codewords.size() == (size_t)rounds && res.roots.size() == (size_t)rounds && seen.size() == (size_t)rounds);
  // oracle replay of the loop
  std::vector<uint64_t> cur = raw, om(omega.value.begin(), omega.value.end()), of(offset.value.begin(), offset.value.end());
  size_t len = n;
  for (int r = 0; r < rounds; r++) {
    std::vector<uint8_t> leaves(len * 25), root(48);
    std::vector<uint64_t> off(len + 1);
    orc_bincode_field_vector(cur.data(), 2, len, leaves.data(), off.data());
    size_t rl = 0;
    orc_merkle_commit_ref(leaves.data(), off.data(), len, root.data(), &rl);
    root.resize(rl);
    CHECK(res.roots[r] == root && seen[r] == root);
    CHECK(res.codewords[r].size() == len);
    bool same = true;
    for (size_t i = 0; i < len; i++) same = same && res.codewords[r][i].value[0] == cur[2 * i] && res.codewords[r][i].value[1] == cur[2 * i + 1];
    CHECK(same);
    if (r == rounds - 1) break;
    uint64_t alpha[2] = {0x1234567ULL * (r + 1) + root[0], root[1]};
    std::vector<uint64_t> nxt(len);   // (len/2) elements x 2 limbs
    orc_fri_fold_ref(MZK_FIELD_M128, cur.data(), len, alpha, of.data(), om.data(), nxt.data());
    uint64_t t[2];
    orc_field_mul(MZK_FIELD_M128, om.data(), om.data(), t); om.assign(t, t + 2);
    orc_field_mul(MZK_FIELD_M128, of.data(), of.data(), t); of.assign(t, t + 2);
    cur = nxt;
    len /= 2;
  }
}

static void test_g2() {   // bn128.rs:306-323 through the MSM, plus the G2 side of batch_verify_kzg (kzg.rs:110-114)
  const uint64_t gen[16] = {   // BN128::generator_g2(), bn128.rs:190-206
      0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL,
      0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL,
      0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL,
      0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
  G2Point g2 = G2Point::from_wire(gen);
  auto msm = [&](std::vector<uint64_t> ks) {
    Polynomial<FqOrder> f;
    std::vector<G2Point> pw;
    for (auto k : ks) { f.coef.push_back(FqOrder::from_value(k)); pw.push_back(g2); }
    return eval_with_powers_on_curve_g2(f, pw);
  };
  CHECK(msm({2, 1, 1}) == msm({4}));
  CHECK(msm({9, 5}) == msm({12, 2}));
  uint64_t four[4] = {4, 0, 0, 0}, w[16];
  orc_g2_mul(gen, four, 4, w);
  CHECK(msm({4}) == G2Point::from_wire(w));
  // powers_2 for alpha = 7: [1, 7, 49] g2; z = (X - 3)(X - 5) = 15 - 8X + X^2 -> z(7) g2 = 8 g2
  auto p2 = setup_kzg_powers_2_with_alpha(g2, FqOrder::from_value(7), 2);
  uint64_t k49[4] = {49, 0, 0, 0};
  orc_g2_mul(gen, k49, 4, w);
  CHECK(p2.size() == 3 && p2[0] == g2 && p2[2] == G2Point::from_wire(w));
  Polynomial<FqOrder> z;
  z.coef = {FqOrder::from_value(15), FqOrder::from_value(0), FqOrder::from_value(1)};
  // -8 mod r
  const uint64_t rm8[4] = {0x43e1f593f0000001ULL - 8, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  z.coef[1] = FqOrder::from_limbs(rm8);
  uint64_t k8[4] = {8, 0, 0, 0};
  orc_g2_mul(gen, k8, 4, w);
  CHECK(eval_with_powers_on_curve_g2(z, p2) == G2Point::from_wire(w));
  // zksnark/utils.rs:83-92 accumulate_curve_points: zip semantics (the shorter length wins), G1 and G2
  {
    std::vector<G2Point> gv{g2, g2, g2, g2};
    std::vector<FqOrder> as{FqOrder::from_value(3), FqOrder::from_value(5)};          // 2 weights, 4 points -> 8 g2
    CHECK(accumulate_curve_points(gv, as) == G2Point::from_wire(w));
    auto g1 = BN128::generator_g1();
    std::vector<G1Point> hv{g1, g1};
    std::vector<FqOrder> bs{FqOrder::from_value(2), FqOrder::from_value(4), FqOrder::from_value(100)};   // 3 weights, 2 points -> 6 g1
    uint64_t g1w[8], k6[4] = {6, 0, 0, 0}, want6[8];
    g1.to_wire(g1w);
    orc_ec_mul(0, g1w, k6, 4, want6);
    CHECK(accumulate_curve_points(hv, bs) == G1Point::from_wire(want6));
    CHECK(accumulate_curve_points(std::vector<G1Point>{}, bs) == G1Point::from_wire(std::vector<uint64_t>(8, 0).data()));   // empty -> infinity
  }
}


// the batch:: extensions against the loops over single calls they stand for
static void test_batch_extensions() {
  typedef FiniteFieldElement<M128> F;
  const unsigned logn = 9;
  const size_t n = 1u << logn, rows = 5;
  std::vector<std::vector<F>> regs(rows, std::vector<F>(n));
  std::vector<uint64_t> raw(2 * n * rows);
  orc_synth_vector(MZK_FIELD_M128, 77, n * rows, raw.data(), 1);
  for (size_t r = 0; r < rows; r++)
    for (size_t i = 0; i < n; i++) regs[r][i] = F::from_limbs(&raw[2 * (r * n + i)]);
  F root = get_nth_root_of_m128(logn);
  auto fwd = batch::ntt(root, regs);
  auto back = batch::ntt(root, fwd, true);
  for (size_t r = 0; r < rows; r++) {
    CHECK(fwd[r] == ntt(root, regs[r]));
    CHECK(back[r] == regs[r]);
  }
  // Polynomial::scale (polynomial.rs:167-174) and the transform through the multi-context entry point (one context here)
  {
    Polynomial<F> pl{regs[0]};
    auto sc = pl.scale(F::from_value(3));
    uint64_t pw[2] = {1, 0}, three[2] = {3, 0};
    bool ok = sc.coef.size() == n;
    for (size_t i = 0; i < n && ok; i++) {
      uint64_t want[2], nx[2];
      orc_field_mul(MZK_FIELD_M128, regs[0][i].value.data(), pw, want);
      ok = sc.coef[i] == F::from_limbs(want);
      orc_field_mul(MZK_FIELD_M128, pw, three, nx);
      pw[0] = nx[0]; pw[1] = nx[1];
    }
    CHECK(ok);
    CHECK(batch::ntt_multi(root, regs[1]) == fwd[1]);
    CHECK(batch::ntt_multi(root, fwd[1], true) == regs[1]);
  }
  // low-degree extension of all registers, then one Merkle root per codeword (fast_stark.rs:231-243)
  const size_t order = 4 * n;
  F big = get_nth_root_of_m128(logn + 2), offset = F::from_value(3);
  std::vector<Polynomial<F>> polys;
  for (auto& r : regs) polys.push_back(Polynomial<F>{r});
  auto lde = batch::fast_coset_evaluate(polys, offset, big, order);
  auto roots = batch::commit_codewords(lde);
  CHECK(lde.size() == rows && roots.size() == rows);
  for (size_t r = 0; r < rows; r++) {
    CHECK(lde[r] == fast_coset_evaluate(polys[r], offset, big, order));
    CHECK(roots[r] == commit_codeword(lde[r]));
  }
  // openings of one codeword: same paths as Merkle::open on the serialized leaves, and they verify
  {
    auto w = to_wire(lde[1]);
    std::vector<uint8_t> blob(order * 25);
    std::vector<uint64_t> off(order + 1);
    orc_bincode_field_vector(w.data(), 2, order, blob.data(), off.data());
    std::vector<std::vector<uint8_t>> leafs;
    for (size_t i = 0; i < order; i++) leafs.emplace_back(blob.begin() + off[i], blob.begin() + off[i + 1]);
    std::vector<size_t> idx{0, 5, order / 2 + 5, order - 1, 5};
    auto paths = batch::open_codeword(idx, lde[1]);
    CHECK(paths.size() == idx.size());
    for (size_t q = 0; q < idx.size(); q++) {
      CHECK(paths[q] == Merkle::open(idx[q], leafs));
      CHECK(oracle_verify(roots[1], idx[q], paths[q], leafs[idx[q]]));
    }
  }
  // interpolation of all registers over the trace domain (fast_stark.rs:203-215)
  {
    std::vector<F> domain(n);
    uint64_t x[2] = {1, 0};
    for (size_t i = 0; i < n; i++) { domain[i] = F::from_limbs(x); uint64_t y[2]; orc_field_mul(MZK_FIELD_M128, x, root.value.data(), y); x[0] = y[0]; x[1] = y[1]; }
    F big3 = get_nth_root_of_m128(logn + 1);
    auto ps = batch::fast_interpolate(domain, regs, big3, 2 * n);
    auto d = to_wire(domain);
    for (size_t r = 0; r < rows; r++) {
      auto v = to_wire(regs[r]);
      std::vector<uint64_t> out(2 * n);
      size_t len = 0;
      expect(mzk_fast_interpolate(MZK_FIELD_M128, d.data(), v.data(), n, big3.value.data(), 2 * n, out.data(), &len));
      out.resize(2 * len);
      CHECK(ps[r].coef.size() == len && to_wire(ps[r].coef) == out);
      CHECK(ps[r].coef == intt(root, regs[r]));   // on the subgroup itself interpolation is the inverse transform
    }
  }
  // commitments against a resident SRS
  {
    auto g1 = BN128::generator_g1();
    const size_t deg = 300;
    auto pk = setup_kzg_with_alpha(g1, FqOrder::from_value(7), deg);
    std::vector<uint64_t> sc(4 * (deg + 1) * 3);
    orc_synth_vector(MZK_FIELD_FR, 9, (deg + 1) * 3, sc.data(), 1);
    std::vector<Polynomial<FqOrder>> fs(3);
    const size_t lens[3] = {deg + 1, 17, 1};
    for (size_t k = 0; k < 3; k++)
      for (size_t i = 0; i < lens[k]; i++) fs[k].coef.push_back(FqOrder::from_limbs(&sc[4 * (k * (deg + 1) + i)]));
    batch::SrsHandle srs(pk);
    auto cs = batch::commit_kzg(fs, srs);
    CHECK(cs.size() == 3);
    for (size_t k = 0; k < 3; k++) CHECK(cs[k] == commit_kzg(fs[k], pk));
    CHECK(batch::commit_kzg({}, srs).empty());
    // six polynomials: the grid-batched pass (mzk_kzg_commit_srs_batch routes there from four on), over the window tables and
    // over direct tables (mzk_srs_build_direct); openings per polynomial at its own point (das/avail.rs:132) the same way
    std::vector<Polynomial<FqOrder>> gs(6);
    std::vector<FqOrder> us;
    for (size_t k = 0; k < 6; k++) {
      const size_t len = k == 5 ? 1 : deg + 1 - 13 * k;
      for (size_t i = 0; i < len; i++) gs[k].coef.push_back(FqOrder::from_limbs(&sc[4 * ((k % 3) * (deg + 1) + i + k)]));
      us.push_back(FqOrder::from_value(1000 + 7 * k));
    }
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 1) CHECK(srs.build_direct(9) == 9);
      auto ds = batch::commit_kzg(gs, srs);
      auto ps = batch::open_kzg(gs, us, srs);
      CHECK(ds.size() == 6 && ps.size() == 6);
      for (size_t k = 0; k < 6; k++) {
        CHECK(ds[k] == commit_kzg(gs[k], pk));
        auto pr = open_kzg(gs[k], us[k], pk);
        CHECK(ps[k].y == pr.y && ps[k].w == pr.w);
      }
    }
  }
}

int main() {
  expect(mzk_init(0));
  test_ntt();
  test_kzg();
  test_batch_kzg();
  test_g1();
  test_fast_multiply();
  test_panics();
  test_merkle();
  test_fri_commit();
  test_g2();
  test_batch_extensions();
  mzk_shutdown();
  if (failures) { printf("%d check(s) failed\n", failures); return 1; }
  printf("all reference-style tests passed\n");
  return 0;
}
