// myzkp.hpp -- host-side mirror of MyZKP's Rust surface for the MSM / NTT path, over the C ABI.
//
// The reference is Rust and rustc is not available in this environment, so the host side above
// include/mzk.h is written in C++ with the SAME names, argument meaning and error behaviour as the
// Rust items it mirrors (paths relative to myzkp/src/modules/):
//
//   FiniteFieldElement<M>, ModulusValue      algebra/field.rs:87-133, :94-96
//   EllipticCurvePoint (BN254 G1 only)       algebra/curve/curve.rs:17-46, curve/bn128.rs:31
//   Polynomial<F>                            algebra/polynomial.rs:69-74
//     ::eval_with_powers_on_curve            polynomial.rs:156-165      -> mzk_msm_g1_bn254
//     ::fft_multiply                         polynomial.rs:242-276      -> mzk_fft_multiply
//     ::scale                                polynomial.rs:167-174      -> mzk_poly_scale
//   ntt / intt / fast_multiply / fast_coset_evaluate   algebra/ntt.rs:7-116, :254-269
//   PublicKeyKZG, ProofKZG, setup_kzg_with_alpha / commit_kzg / open_kzg   algebra/kzg.rs:8-72
//   get_nth_root_of_m128                     zkstark/fri.rs:423-447
//   fast_coset_divide                        algebra/ntt.rs:271-330
//   G2Point, eval_with_powers_on_curve_g2, setup_kzg_powers_2_with_alpha   curve/bn128.rs:33-49, algebra/kzg.rs:42-55,114
//   accumulate_curve_points (G1 and G2)     zksnark/utils.rs:83-92
//   Merkle::commit / open, commit_codeword   algebra/merkle.rs:15-46, zkstark/fri.rs:160-166
//   fri_split_and_fold, fri_commit           zkstark/fri.rs:144-209
//
// Values are held canonical (u64 limbs) -- the ABI wire format; arithmetic on single elements that the
// reference does on the host (a handful of scalar ops in tests) is not offered here: this header only
// marshals the bulk operations to the GPU.  Rust panics become C++ exceptions carrying the same text.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>
#include <exception>
#include "../../include/mzk.h"

namespace myzkp {

struct Panic : std::runtime_error {
  int code;
  Panic(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void expect(int rc) {
  if (rc != MZK_OK) throw Panic(rc, mzk_last_error());
}
// scratch memory of the library (no reference counterpart: the reference's Vec buffers die with each call): what a context may keep
// between calls, and an explicit release; MZK_E_NOMEM surfaces through expect() like every other status
inline void set_workspace_budget(size_t bytes) { expect(mzk_set_workspace_budget(bytes)); }
inline size_t trim_workspace() { size_t r = 0; expect(mzk_trim_workspace(&r)); return r; }

// ---- ModulusValue phantom types (field.rs:94-96, :408-431; bn128.rs:19-30; fri.rs:408) -------------
struct ModEIP197 { static constexpr int FIELD_ID = MZK_FIELD_FR; static constexpr int LIMBS = 4; };
struct BN128Modulus { static constexpr int FIELD_ID = MZK_FIELD_FQ; static constexpr int LIMBS = 4; };
struct M128 { static constexpr int FIELD_ID = MZK_FIELD_M128; static constexpr int LIMBS = 2; };

template <class M> struct FiniteFieldElement {
  std::array<uint64_t, M::LIMBS> value{};  // canonical little-endian limbs (sanitize()d, field.rs:260-270)
  FiniteFieldElement() = default;
  static FiniteFieldElement from_value(uint64_t v) {  // Ring::from_value for small values
    FiniteFieldElement e;
    e.value[0] = v;
    return e;
  }
  static FiniteFieldElement from_limbs(const uint64_t* l) {
    FiniteFieldElement e;
    std::memcpy(e.value.data(), l, 8 * M::LIMBS);
    return e;
  }
  static FiniteFieldElement zero() { return FiniteFieldElement(); }
  static FiniteFieldElement one() { return from_value(1); }
  bool is_zero() const {
    for (auto x : value) if (x) return false;
    return true;
  }
  bool operator==(const FiniteFieldElement& o) const { return value == o.value; }
  bool operator!=(const FiniteFieldElement& o) const { return !(*this == o); }
};
using FqOrder = FiniteFieldElement<ModEIP197>;  // bn128.rs:30
using Fq = FiniteFieldElement<BN128Modulus>;    // bn128.rs:29

// ---- EllipticCurvePoint<Fq, BN128Curve> = G1Point (curve.rs:17-46; bn128.rs:31) ------------------------
struct G1Point {
  bool infinity = true;  // x, y = None in the reference
  Fq x, y;
  static G1Point point_at_infinity() { return G1Point(); }
  static G1Point new_point(const Fq& x, const Fq& y) {
    G1Point p;
    p.infinity = false; p.x = x; p.y = y;
    return p;
  }
  bool is_point_at_infinity() const { return infinity; }
  bool operator==(const G1Point& o) const { return infinity == o.infinity && (infinity || (x == o.x && y == o.y)); }
  void to_wire(uint64_t* w) const {  // all-zero = infinity
    if (infinity) { std::memset(w, 0, 64); return; }
    std::memcpy(w, x.value.data(), 32);
    std::memcpy(w + 4, y.value.data(), 32);
  }
  static G1Point from_wire(const uint64_t* w) {
    bool z = true;
    for (int i = 0; i < 8; i++) z = z && w[i] == 0;
    if (z) return point_at_infinity();
    return new_point(Fq::from_limbs(w), Fq::from_limbs(w + 4));
  }
};
struct BN128 {  // bn128.rs:183-216
  static G1Point generator_g1() { return G1Point::new_point(Fq::from_value(1), Fq::from_value(2)); }
};

template <class F> static std::vector<uint64_t> to_wire(const std::vector<F>& v) {
  std::vector<uint64_t> w(v.size() * F().value.size());
  for (size_t i = 0; i < v.size(); i++) std::memcpy(&w[i * v[i].value.size()], v[i].value.data(), 8 * v[i].value.size());
  return w;
}
template <class F> static std::vector<F> from_wire(const std::vector<uint64_t>& w, size_t n) {
  std::vector<F> v(n);
  for (size_t i = 0; i < n; i++) std::memcpy(v[i].value.data(), &w[i * v[i].value.size()], 8 * v[i].value.size());
  return v;
}
static inline std::vector<uint64_t> points_to_wire(const std::vector<G1Point>& p) {
  std::vector<uint64_t> w(p.size() * 8);
  for (size_t i = 0; i < p.size(); i++) p[i].to_wire(&w[8 * i]);
  return w;
}

// ---- Polynomial<F> (polynomial.rs:69-74), coefficients in ascending degree ---------------------------
template <class F> struct Polynomial {
  std::vector<F> coef;

  // polynomial.rs:156-165.  `powers.len() < coef.len()` is the reference's slice-index panic.
  G1Point eval_with_powers_on_curve(const std::vector<G1Point>& powers) const {
    static_assert(std::is_same<F, FqOrder>::value, "MSM scalars live in FqOrder");
    if (powers.size() < coef.size())
      throw Panic(MZK_E_LENGTH, "index out of bounds: the len is " + std::to_string(powers.size()) + " but the index is " + std::to_string(powers.size()));
    auto s = to_wire(coef);
    std::vector<uint64_t> p(coef.size() * 8);
    for (size_t i = 0; i < coef.size(); i++) powers[i].to_wire(&p[8 * i]);
    uint64_t out[8];
    expect(mzk_msm_g1_bn254(s.data(), p.data(), coef.size(), out));
    return G1Point::from_wire(out);
  }

  // polynomial.rs:242-276
  Polynomial fft_multiply(const Polynomial& other, const F& omega) const {
    auto a = to_wire(coef), b = to_wire(other.coef);
    std::vector<uint64_t> out((coef.size() + other.coef.size() + 1) * F().value.size());
    size_t n = 0;
    expect(mzk_fft_multiply(field_id(), a.data(), coef.size(), b.data(), other.coef.size(), omega.value.data(), out.data(), &n));
    return Polynomial{from_wire<F>(out, n)};
  }
  // polynomial.rs:167-174: coef[i] * offset^i
  Polynomial scale(const F& offset) const {
    auto a = to_wire(coef);
    std::vector<uint64_t> out(a.size());
    expect(mzk_poly_scale(field_id(), a.data(), coef.size(), offset.value.data(), nullptr, out.data()));
    return Polynomial{from_wire<F>(out, coef.size())};
  }
  static int field_id() { return F().value.size() == 2 ? MZK_FIELD_M128 : MZK_FIELD_FR; }
};

// ---- algebra/ntt.rs ---------------------------------------------------------------------------------------
template <class F> std::vector<F> ntt(const F& primitive_root, const std::vector<F>& values) {  // ntt.rs:7-48
  auto in = to_wire(values);
  std::vector<uint64_t> out(in.size());
  expect(mzk_ntt(Polynomial<F>::field_id(), primitive_root.value.data(), in.data(), out.data(), values.size(), 0));
  return from_wire<F>(out, values.size());
}
template <class F> std::vector<F> intt(const F& primitive_root, const std::vector<F>& values) {  // ntt.rs:50-64
  auto in = to_wire(values);
  std::vector<uint64_t> out(in.size());
  expect(mzk_ntt(Polynomial<F>::field_id(), primitive_root.value.data(), in.data(), out.data(), values.size(), 1));
  return from_wire<F>(out, values.size());
}
template <class F>
Polynomial<F> fast_multiply(const Polynomial<F>& lhs, const Polynomial<F>& rhs, const F& primitive_root, size_t root_order) {  // ntt.rs:66-116
  auto a = to_wire(lhs.coef), b = to_wire(rhs.coef);
  const size_t cap = std::max(root_order, lhs.coef.size() + rhs.coef.size()) + 1;
  std::vector<uint64_t> out(cap * F().value.size());
  size_t n = 0;
  expect(mzk_fast_multiply(Polynomial<F>::field_id(), a.data(), lhs.coef.size(), b.data(), rhs.coef.size(),
                           primitive_root.value.data(), root_order, out.data(), &n));
  return Polynomial<F>{from_wire<F>(out, n)};
}
template <class F>
std::vector<F> fast_coset_evaluate(const Polynomial<F>& polynomial, const F& offset, const F& generator, size_t order) {  // ntt.rs:254-269
  auto c = to_wire(polynomial.coef);
  std::vector<uint64_t> out(order * F().value.size());
  expect(mzk_coset_lde(Polynomial<F>::field_id(), c.data(), polynomial.coef.size(), offset.value.data(), generator.value.data(),
                       out.data(), order));
  return from_wire<F>(out, order);
}
// zkstark/fri.rs:423-447 (n given as log2)
inline FiniteFieldElement<M128> get_nth_root_of_m128(unsigned log2_n) {
  FiniteFieldElement<M128> r;
  expect(mzk_root_of_unity(MZK_FIELD_M128, log2_n, r.value.data()));
  return r;
}
inline FqOrder get_nth_root_of_fr(unsigned log2_n) {  // the reference ships none
  FqOrder r;
  expect(mzk_root_of_unity(MZK_FIELD_FR, log2_n, r.value.data()));
  return r;
}

// ---- algebra/kzg.rs ---------------------------------------------------------------------------------------
struct PublicKeyKZG { std::vector<G1Point> powers_1; };  // kzg.rs:8-11 (powers_2 lives in G2: out of scope)
using CommitmentKZG = G1Point;                            // kzg.rs:13
struct ProofKZG { FqOrder y; G1Point w; };                // kzg.rs:15-18

// setup_kzg (kzg.rs:27-40) with the trapdoor supplied (the reference draws it from thread_rng)
inline PublicKeyKZG setup_kzg_with_alpha(const G1Point& g1, const FqOrder& alpha, size_t max_d) {
  uint64_t g[8];
  g1.to_wire(g);
  std::vector<uint64_t> out((max_d + 1) * 8);
  expect(mzk_kzg_setup_g1(alpha.value.data(), g, max_d, out.data()));
  PublicKeyKZG pk;
  pk.powers_1.resize(max_d + 1);
  for (size_t i = 0; i <= max_d; i++) pk.powers_1[i] = G1Point::from_wire(&out[8 * i]);
  return pk;
}
inline CommitmentKZG commit_kzg(const Polynomial<FqOrder>& f, const PublicKeyKZG& pk) {  // kzg.rs:57-59
  return f.eval_with_powers_on_curve(pk.powers_1);
}
inline ProofKZG open_kzg(const Polynomial<FqOrder>& f, const FqOrder& u, const PublicKeyKZG& pk) {  // kzg.rs:61-72
  if (f.coef.size() > 1 && pk.powers_1.size() < f.coef.size() - 1)
    throw Panic(MZK_E_LENGTH, "index out of bounds: the len is " + std::to_string(pk.powers_1.size()) + " but the index is " + std::to_string(pk.powers_1.size()));
  auto c = to_wire(f.coef);
  auto p = points_to_wire(pk.powers_1);
  ProofKZG pr;
  uint64_t w[8];
  expect(mzk_kzg_open(c.data(), f.coef.size(), u.value.data(), p.data(), pr.y.value.data(), w));
  pr.w = G1Point::from_wire(w);
  return pr;
}

struct BatchProofKZG { std::vector<FqOrder> ys; G1Point w; };  // kzg.rs:20-23
using ProofDegreeBound = G1Point;                              // kzg.rs:25
inline BatchProofKZG batch_open_kzg(const Polynomial<FqOrder>& f, const std::vector<FqOrder>& us, const PublicKeyKZG& pk) {  // kzg.rs:74-88
  const size_t nq = f.coef.size() > us.size() ? f.coef.size() - us.size() : 0;
  if (pk.powers_1.size() < nq)
    throw Panic(MZK_E_LENGTH, "index out of bounds: the len is " + std::to_string(pk.powers_1.size()) + " but the index is " + std::to_string(pk.powers_1.size()));
  auto c = to_wire(f.coef), u = to_wire(us);
  auto p = points_to_wire(pk.powers_1);
  std::vector<uint64_t> ys(us.size() * 4 + 4);
  uint64_t w[8];
  expect(mzk_kzg_batch_open(c.data(), f.coef.size(), u.data(), us.size(), p.data(), ys.data(), w));
  BatchProofKZG pr;
  pr.ys = from_wire<FqOrder>(ys, us.size());
  pr.w = G1Point::from_wire(w);
  return pr;
}
inline ProofDegreeBound prove_degree_bound(const Polynomial<FqOrder>& f, const PublicKeyKZG& pk, size_t d) {  // kzg.rs:121-134
  auto c = to_wire(f.coef);
  auto p = points_to_wire(pk.powers_1);
  uint64_t w[8];
  expect(mzk_kzg_prove_degree_bound(c.data(), f.coef.size(), p.data(), pk.powers_1.size(), d, w));
  return G1Point::from_wire(w);
}
// ntt::fast_coset_divide (ntt.rs:271-330)
template <class F>
Polynomial<F> fast_coset_divide(const Polynomial<F>& lhs, const Polynomial<F>& rhs, const F& offset, const F& primitive_root, size_t root_order) {
  auto a = to_wire(lhs.coef), b = to_wire(rhs.coef);
  std::vector<uint64_t> out((lhs.coef.size() + 1) * F().value.size());
  size_t len = 0;
  expect(mzk_fast_coset_divide(Polynomial<F>::field_id(), a.data(), lhs.coef.size(), b.data(), rhs.coef.size(), offset.value.data(),
                               primitive_root.value.data(), root_order, out.data(), &len));
  return Polynomial<F>{from_wire<F>(out, len)};
}
// one round of the split-and-fold in FRI::commit (zkstark/fri.rs:182-193)
template <class F>
std::vector<F> fri_split_and_fold(const std::vector<F>& codeword, const F& alpha, const F& offset, const F& omega) {
  auto c = to_wire(codeword);
  std::vector<uint64_t> out((codeword.size() / 2 + 1) * F().value.size());
  expect(mzk_fri_fold(Polynomial<F>::field_id(), c.data(), codeword.size(), alpha.value.data(), offset.value.data(), omega.value.data(), out.data()));
  return from_wire<F>(out, codeword.size() / 2);
}


// ---- G2 (bn128.rs:33-49): Fq2 = Fq[u]/(u^2 + 1), G2Point = EllipticCurvePoint<Fq2, BN128Curve> ----------
struct Fq2 {   // ExtendedFieldElement: poly.coef = [c0, c1]
  Fq c0, c1;
  bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
};
struct G2Point {
  bool infinity = true;
  Fq2 x, y;
  static G2Point point_at_infinity() { return G2Point(); }
  static G2Point new_point(const Fq2& x, const Fq2& y) { G2Point p; p.infinity = false; p.x = x; p.y = y; return p; }
  bool is_point_at_infinity() const { return infinity; }
  bool operator==(const G2Point& o) const { return infinity == o.infinity && (infinity || (x == o.x && y == o.y)); }
  void to_wire(uint64_t* w) const {
    if (infinity) { std::memset(w, 0, 128); return; }
    std::memcpy(w, x.c0.value.data(), 32); std::memcpy(w + 4, x.c1.value.data(), 32);
    std::memcpy(w + 8, y.c0.value.data(), 32); std::memcpy(w + 12, y.c1.value.data(), 32);
  }
  static G2Point from_wire(const uint64_t* w) {
    bool z = true;
    for (int i = 0; i < 16; i++) z = z && w[i] == 0;
    if (z) return point_at_infinity();
    return new_point(Fq2{Fq::from_limbs(w), Fq::from_limbs(w + 4)}, Fq2{Fq::from_limbs(w + 8), Fq::from_limbs(w + 12)});
  }
};
// Polynomial::eval_with_powers_on_curve over G2 (polynomial.rs:156-165 as called at kzg.rs:114)
inline G2Point eval_with_powers_on_curve_g2(const Polynomial<FqOrder>& f, const std::vector<G2Point>& powers) {
  if (powers.size() < f.coef.size()) throw Panic(MZK_E_LENGTH, "index out of bounds: powers.len() < coef.len() (polynomial.rs:162)");
  auto s = to_wire(f.coef);
  std::vector<uint64_t> p(16 * f.coef.size() + 16), out(16);
  for (size_t i = 0; i < f.coef.size(); i++) powers[i].to_wire(&p[16 * i]);
  expect(mzk_msm_g2_bn254(s.data(), p.data(), f.coef.size(), out.data()));
  return G2Point::from_wire(out.data());
}
// zksnark/utils.rs:83-92 accumulate_curve_points -- the reference's second spelling of the MSM: sum_i g_vec[i] * assignment[i]
// over zip(g_vec, assignment), i.e. the SHORTER of the two lengths (no index panic), G1 or G2 points
inline G1Point accumulate_curve_points(const std::vector<G1Point>& g_vec, const std::vector<FqOrder>& assignment) {
  const size_t n = g_vec.size() < assignment.size() ? g_vec.size() : assignment.size();
  std::vector<uint64_t> s(4 * n + 4), p(8 * n + 8);
  for (size_t i = 0; i < n; i++) { std::memcpy(&s[4 * i], assignment[i].value.data(), 32); g_vec[i].to_wire(&p[8 * i]); }
  uint64_t out[8];
  expect(mzk_msm_g1_bn254(s.data(), p.data(), n, out));
  return G1Point::from_wire(out);
}
inline G2Point accumulate_curve_points(const std::vector<G2Point>& g_vec, const std::vector<FqOrder>& assignment) {
  const size_t n = g_vec.size() < assignment.size() ? g_vec.size() : assignment.size();
  std::vector<uint64_t> s(4 * n + 4), p(16 * n + 16), out(16);
  for (size_t i = 0; i < n; i++) { std::memcpy(&s[4 * i], assignment[i].value.data(), 32); g_vec[i].to_wire(&p[16 * i]); }
  expect(mzk_msm_g2_bn254(s.data(), p.data(), n, out.data()));
  return G2Point::from_wire(out.data());
}
// powers_2 of setup_kzg_with_full_g2 (kzg.rs:42-55) for a caller-supplied alpha
inline std::vector<G2Point> setup_kzg_powers_2_with_alpha(const G2Point& g2, const FqOrder& alpha, size_t max_d) {
  uint64_t g[16];
  g2.to_wire(g);
  std::vector<uint64_t> w(16 * (max_d + 1));
  expect(mzk_kzg_setup_g2(alpha.value.data(), g, max_d, w.data()));
  std::vector<G2Point> out;
  for (size_t i = 0; i <= max_d; i++) out.push_back(G2Point::from_wire(&w[16 * i]));
  return out;
}

// ---- Merkle (algebra/merkle.rs) and the FRI commit phase (zkstark/fri.rs:144-209) ------------------------
typedef std::vector<uint8_t> MerkleRoot;               // merkle.rs:4
typedef std::vector<std::vector<uint8_t>> MerklePath;  // merkle.rs:5
struct Merkle {
  static void flatten(const std::vector<std::vector<uint8_t>>& leafs, std::vector<uint8_t>& blob, std::vector<uint64_t>& off, size_t& stride) {
    off.assign(1, 0);
    stride = 32;
    for (auto& l : leafs) { blob.insert(blob.end(), l.begin(), l.end()); off.push_back(blob.size()); if (l.size() > stride) stride = l.size(); }
  }
  static MerkleRoot commit(const std::vector<std::vector<uint8_t>>& leafs) {   // merkle.rs:15-25
    std::vector<uint8_t> blob; std::vector<uint64_t> off; size_t stride;
    flatten(leafs, blob, off, stride);
    MerkleRoot root(stride);
    size_t len = 0;
    expect(mzk_merkle_commit_bytes(blob.data(), off.data(), leafs.size(), root.data(), root.size(), &len));
    root.resize(len);
    return root;
  }
  static MerklePath open(size_t index, const std::vector<std::vector<uint8_t>>& leafs) {   // merkle.rs:28-46
    std::vector<uint8_t> blob; std::vector<uint64_t> off; size_t stride;
    flatten(leafs, blob, off, stride);
    mzk_merkle* t = nullptr;
    expect(mzk_merkle_build_bytes(blob.data(), off.data(), leafs.size(), &t));
    std::vector<uint8_t> buf(stride * 64);
    uint64_t lens[64];
    size_t depth = 0;
    const int rc = mzk_merkle_open(t, index, buf.data(), stride, lens, &depth);
    mzk_merkle_free(t);
    expect(rc);
    MerklePath path;
    for (size_t k = 0; k < depth; k++) path.emplace_back(buf.begin() + k * stride, buf.begin() + k * stride + lens[k]);
    return path;
  }
};
// Merkle::commit(&codeword.map(|c| bincode::serialize(&c))) as the provers write it (fri.rs:160-166)
template <class F> MerkleRoot commit_codeword(const std::vector<F>& codeword) {
  auto c = to_wire(codeword);
  MerkleRoot root(48);
  size_t len = 0;
  expect(mzk_merkle_commit_field(Polynomial<F>::field_id(), c.data(), codeword.size(), root.data(), root.size(), &len));
  root.resize(len);
  return root;
}
// FRI::commit (fri.rs:144-209): `challenge(round, last, root) -> alpha` owns the proof stream (push the root;
// sample alpha unless last).  Returns (codewords, roots) like the reference.
template <class F> struct FriCommitment { std::vector<std::vector<F>> codewords; std::vector<MerkleRoot> roots; };
template <class F, class Fn>
FriCommitment<F> fri_commit(const std::vector<F>& initial_codeword, const F& omega, const F& offset, int num_rounds, Fn&& challenge) {
  typedef typename std::remove_reference<Fn>::type Callee;
  // a throwing transcript must not unwind through the C library: report failure (-> MZK_E_CALLBACK), rethrow below
  struct Ctx { Callee* fn; std::exception_ptr err; } cx{&challenge, nullptr};
  auto tramp = [](void* user, int round, int last, const uint8_t* root, size_t root_len, uint64_t* alpha_out) -> int {
    Ctx* c = static_cast<Ctx*>(user);
    try {
      F a = (*c->fn)(round, last != 0, MerkleRoot(root, root + root_len));
      std::memcpy(alpha_out, a.value.data(), 8 * a.value.size());
      return 0;
    } catch (...) {
      c->err = std::current_exception();
      return 1;
    }
  };
  const size_t n = initial_codeword.size(), nl = F().value.size();
  size_t total = 0;
  for (int r = 0; r < num_rounds; r++) total += n >> r;
  auto c = to_wire(initial_codeword);
  std::vector<uint8_t> roots(48 * (size_t)(num_rounds > 0 ? num_rounds : 0) + 1);
  std::vector<uint64_t> lens(num_rounds > 0 ? num_rounds : 1), all((total + 1) * nl);
  const int frc = mzk_fri_commit(Polynomial<F>::field_id(), c.data(), n, omega.value.data(), offset.value.data(), num_rounds, +tramp, &cx, roots.data(),
                                 lens.data(), all.data());
  if (cx.err) std::rethrow_exception(cx.err);
  expect(frc);
  FriCommitment<F> out;
  size_t at = 0;
  for (int r = 0; r < num_rounds; r++) {
    std::vector<uint64_t> w(all.begin() + at * nl, all.begin() + (at + (n >> r)) * nl);
    out.codewords.push_back(from_wire<F>(w, n >> r));
    out.roots.emplace_back(roots.begin() + 48 * r, roots.begin() + 48 * r + lens[r]);
    at += n >> r;
  }
  return out;
}

// ---- extensions: the reference's per-item loops as one call ------------------------------------------------------
// The reference has no batch forms; its callers loop (fast_stark.rs:231-243 per register, fri.rs:211-260 per query,
// kzg.rs:57-59 per polynomial).  Each function below returns exactly what that loop over the single call returns.
namespace batch {

// PublicKeyKZG.powers_1 kept on the GPU (with its window tables) across commitments
struct SrsHandle {
  mzk_srs* h = nullptr;
  explicit SrsHandle(const PublicKeyKZG& pk) {
    auto p = points_to_wire(pk.powers_1);
    expect(mzk_srs_upload(p.data(), pk.powers_1.size(), &h));
  }
  // every multiple of every window-table row, for batches of short polynomials (mzk_srs_build_direct); returns the width built
  int build_direct(int window_bits = 0, size_t max_bytes = 0) {
    expect(mzk_srs_build_direct(h, window_bits, max_bytes, nullptr));
    return mzk_srs_direct_bits(h);
  }
  SrsHandle(const SrsHandle&) = delete;
  SrsHandle& operator=(const SrsHandle&) = delete;
  ~SrsHandle() { mzk_srs_free(h); }
};
// for f in fs { commit_kzg(f, pk) }   (kzg.rs:57-59); every polynomial padded with zero coefficients to the longest
inline std::vector<CommitmentKZG> commit_kzg(const std::vector<Polynomial<FqOrder>>& fs, const SrsHandle& srs) {
  size_t n = 0;
  for (auto& f : fs) n = f.coef.size() > n ? f.coef.size() : n;
  std::vector<CommitmentKZG> out(fs.size());
  if (fs.empty() || n == 0) return out;
  std::vector<uint64_t> c(fs.size() * n * 4, 0), xy(fs.size() * 8);
  for (size_t i = 0; i < fs.size(); i++)
    for (size_t j = 0; j < fs[i].coef.size(); j++) std::memcpy(&c[(i * n + j) * 4], fs[i].coef[j].value.data(), 32);
  expect(mzk_kzg_commit_srs_batch(srs.h, c.data(), n, fs.size(), xy.data()));
  for (size_t i = 0; i < fs.size(); i++) out[i] = G1Point::from_wire(&xy[8 * i]);
  return out;
}

// for (f, u) in zip(fs, us) { open_kzg(f, u, pk) }   (kzg.rs:61-72; das/avail.rs:132 opens per cell); polynomials padded with zero
// coefficients to the longest -- the quotient of a padded polynomial has zero leading coefficients, the witness is the same point
inline std::vector<ProofKZG> open_kzg(const std::vector<Polynomial<FqOrder>>& fs, const std::vector<FqOrder>& us, const SrsHandle& srs) {
  if (fs.size() != us.size()) throw Panic(MZK_E_ARG, "batch::open_kzg: one point per polynomial");
  size_t n = 0;
  for (auto& f : fs) n = f.coef.size() > n ? f.coef.size() : n;
  std::vector<ProofKZG> out(fs.size());
  if (fs.empty()) return out;
  std::vector<uint64_t> c(fs.size() * n * 4 + 4, 0), u = to_wire(us), ys(fs.size() * 4), ws(fs.size() * 8);
  for (size_t i = 0; i < fs.size(); i++)
    for (size_t j = 0; j < fs[i].coef.size(); j++) std::memcpy(&c[(i * n + j) * 4], fs[i].coef[j].value.data(), 32);
  expect(mzk_kzg_open_srs_batch(srs.h, c.data(), n, fs.size(), u.data(), ys.data(), ws.data()));
  for (size_t i = 0; i < fs.size(); i++) {
    std::memcpy(out[i].y.value.data(), &ys[4 * i], 32);
    out[i].w = G1Point::from_wire(&ws[8 * i]);
  }
  return out;
}

template <class F> static std::vector<uint64_t> rows_to_wire(const std::vector<std::vector<F>>& rows, size_t n) {
  const size_t nl = F().value.size();
  std::vector<uint64_t> w(rows.size() * n * nl, 0);
  for (size_t i = 0; i < rows.size(); i++) {
    if (rows[i].size() > n) throw Panic(MZK_E_ARG, "batch rows must have one length");
    for (size_t j = 0; j < rows[i].size(); j++) std::memcpy(&w[(i * n + j) * nl], rows[i][j].value.data(), 8 * nl);
  }
  return w;
}
template <class F> static std::vector<std::vector<F>> rows_from_wire(const std::vector<uint64_t>& w, size_t rows, size_t n) {
  const size_t nl = F().value.size();
  std::vector<std::vector<F>> out(rows, std::vector<F>(n));
  for (size_t i = 0; i < rows; i++)
    for (size_t j = 0; j < n; j++) std::memcpy(out[i][j].value.data(), &w[(i * n + j) * nl], 8 * nl);
  return out;
}
// for v in rows { ntt(root, v) } / intt   (ntt.rs:7-64); all rows of one power-of-two length
template <class F> std::vector<std::vector<F>> ntt(const F& primitive_root, const std::vector<std::vector<F>>& rows, bool inverse = false) {
  if (rows.empty()) return {};
  const size_t n = rows[0].size();
  for (auto& r : rows) if (r.size() != n) throw Panic(MZK_E_ARG, "batch rows must have one length");
  auto in = rows_to_wire(rows, n);
  std::vector<uint64_t> out(in.size());
  expect(mzk_ntt_batch(Polynomial<F>::field_id(), primitive_root.value.data(), in.data(), out.data(), n, rows.size(), inverse ? 1 : 0));
  return rows_from_wire<F>(out, rows.size(), n);
}
// for p in polys { fast_coset_evaluate(p, offset, generator, order) }   (ntt.rs:254-269); coefficients zero-padded to
// the longest polynomial (a power of two after padding is the caller's business, as in the reference)
template <class F>
std::vector<std::vector<F>> fast_coset_evaluate(const std::vector<Polynomial<F>>& polys, const F& offset, const F& generator, size_t order) {
  if (polys.empty()) return {};
  size_t n = 1;
  for (auto& p : polys) n = p.coef.size() > n ? p.coef.size() : n;
  std::vector<std::vector<F>> rows;
  for (auto& p : polys) rows.push_back(p.coef);
  auto in = rows_to_wire(rows, n);
  std::vector<uint64_t> out(polys.size() * order * F().value.size());
  expect(mzk_coset_lde_batch(Polynomial<F>::field_id(), in.data(), n, offset.value.data(), generator.value.data(), out.data(), order, polys.size()));
  return rows_from_wire<F>(out, polys.size(), order);
}
// for c in codewords { Merkle::commit(c.map(serialize)) }   (fast_stark.rs:231-243); one power-of-two length >= 2
template <class F> std::vector<MerkleRoot> commit_codewords(const std::vector<std::vector<F>>& codewords) {
  if (codewords.empty()) return {};
  const size_t n = codewords[0].size();
  for (auto& r : codewords) if (r.size() != n) throw Panic(MZK_E_ARG, "batch rows must have one length");
  auto in = rows_to_wire(codewords, n);
  std::vector<uint8_t> roots(32 * codewords.size());
  expect(mzk_merkle_commit_field_batch(Polynomial<F>::field_id(), in.data(), n, codewords.size(), roots.data()));
  std::vector<MerkleRoot> out;
  for (size_t i = 0; i < codewords.size(); i++) out.emplace_back(roots.begin() + 32 * i, roots.begin() + 32 * (i + 1));
  return out;
}
// for i in indices { Merkle::open(i, codeword.map(serialize)) }   (fri.rs:211-260): the tree is hashed once
template <class F> std::vector<MerklePath> open_codeword(const std::vector<size_t>& indices, const std::vector<F>& codeword) {
  auto c = to_wire(codeword);
  mzk_merkle* t = nullptr;
  expect(mzk_merkle_build_field(Polynomial<F>::field_id(), c.data(), codeword.size(), &t));
  const size_t stride = 48;
  std::vector<uint64_t> idx(indices.begin(), indices.end()), lens(indices.size() * 64);
  std::vector<uint8_t> buf(indices.size() * 64 * stride);
  size_t depth = 0;
  const int rc = mzk_merkle_open_batch(t, idx.data(), idx.size(), buf.data(), stride, lens.data(), &depth);
  mzk_merkle_free(t);
  expect(rc);
  std::vector<MerklePath> out(indices.size());
  for (size_t q = 0; q < indices.size(); q++)
    for (size_t l = 0; l < depth; l++) {
      const uint8_t* e = &buf[(q * depth + l) * stride];
      out[q].emplace_back(e, e + lens[q * depth + l]);
    }
  return out;
}
// for v in value_rows { fast_interpolate(domain, v, root, root_order) }   (ntt.rs:211-252): one subproduct tree
template <class F>
std::vector<Polynomial<F>> fast_interpolate(const std::vector<F>& domain, const std::vector<std::vector<F>>& value_rows, const F& primitive_root,
                                            size_t root_order) {
  if (value_rows.empty()) return {};
  const size_t n = domain.size(), nl = F().value.size();
  for (auto& r : value_rows) if (r.size() != n) throw Panic(MZK_E_ARG, "batch rows must have one length");
  auto d = to_wire(domain);
  auto v = rows_to_wire(value_rows, n);
  std::vector<uint64_t> out(value_rows.size() * (n ? n : 1) * nl);
  std::vector<size_t> lens(value_rows.size());
  expect(mzk_fast_interpolate_batch(Polynomial<F>::field_id(), d.data(), v.data(), n, value_rows.size(), primitive_root.value.data(), root_order,
                                    out.data(), lens.data()));
  std::vector<Polynomial<F>> res(value_rows.size());
  for (size_t i = 0; i < value_rows.size(); i++) {
    std::vector<uint64_t> w(out.begin() + i * n * nl, out.begin() + (i * n + lens[i]) * nl);
    res[i].coef = from_wire<F>(w, lens[i]);
  }
  return res;
}


// ntt / intt (ntt.rs:7-64) of ONE vector spread over every context of mzk_init_devices (four-step layout, mzk_ntt_multi)
template <class F> std::vector<F> ntt_multi(const F& primitive_root, const std::vector<F>& values, bool inverse = false) {
  auto in = to_wire(values);
  std::vector<uint64_t> out(in.size());
  expect(mzk_ntt_multi(Polynomial<F>::field_id(), primitive_root.value.data(), in.data(), out.data(), values.size(), inverse ? 1 : 0));
  return from_wire<F>(out, values.size());
}

}  // namespace batch

}  // namespace myzkp
