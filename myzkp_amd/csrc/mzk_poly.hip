// mzk_poly.hip -- subproduct-tree polynomial routines on the device: ntt::fast_zerofier, fast_evaluate,
// fast_interpolate (myzkp/src/modules/algebra/ntt.rs:118-252; FastStark::prove interpolates every trace register with
// fast_interpolate, zkstark/fast_stark.rs:209, and builds its transition zerofier with fast_zerofier, :53).
//
// The reference recurses on slices (half = len / 2) and multiplies with fast_multiply; its remainders are schoolbook
// long divisions (polynomial.rs:371-405), i.e. O(n^2).  The three results are mathematically determined --
//   zerofier    Z(X) = prod_i (X - d_i)
//   evaluate    [f(d_i)]_i
//   interpolate the polynomial of degree < n through (d_i, v_i)   (a repeated point contributes inverse(0) = 0, see below)
// -- so any exact algorithm returns the same canonical coefficients; only the LENGTH of the returned vector is an
// artefact of the recursion, and it is reproduced (fast_multiply's untrimmed `order` on its NTT path, ntt.rs:86-93).
//
// Device algorithm (all O(n log^2 n), every level one batched launch group):
//   * the domain is padded with zeros to N = 2^k >= 64 points: Z_pad = Z * X^pad, so results are read with a shift;
//   * level 0: one wave per 64 points builds its monic degree-64 zerofier by 64 rank-1 updates (k_chunk_zerofier);
//   * level l -> l+1: monic pairs  (X^D + a)(X^D + b) = X^2D + X^D (a + b) + a b : only the low parts are stored, and a b
//     (degree <= 2D - 2) is one cyclic product of size 2D -- batched NTTs (mzk_ntt.hip) over all N / D polynomials;
//   * evaluation is the TRANSPOSED algorithm of Bostan-Lecerf-Schost: t_root = middle product of f with
//     1 / rev(Z_pad) mod X^N (Newton), then down the tree  t_left = (t_node * Z_right)[D .. 2D)  (one cyclic product
//     of size 2D per node, re-using the transformed low parts kept from the way up), and a 64-point finish per wave;
//   * interpolation: weights w_i = v_i / Z'(d_i) (Z' evaluated as above), then the same tree upwards:
//     P_parent = X^D (P_l + P_r) + P_l Z_r' + P_r Z_l';
//   * interpolation over the first n points of a power-of-two subgroup (the trace domain of fast_stark.rs:197-215) needs no tree: one
//     inverse transform after the m = next_pow2(n) - n missing values are filled in (k_prefix_weights and below).
#include <algorithm>
#include <vector>
#include "mzk_common.h"
#include "mzk_field_asm.h"

namespace mzk {

constexpr int CHUNK = 64;        // points per level-0 polynomial = one wave

template <class P> __device__ __forceinline__ Fe<P> pl_load(const u32* __restrict__ g, size_t idx) {
  u32 w[P::NW];
  const uint4* p4 = reinterpret_cast<const uint4*>(g + idx * P::NW);
#pragma unroll
  for (int q = 0; q < P::NW / 4; q++) { uint4 v = p4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
  return fe_unpack<P>(w);
}
template <class P> __device__ __forceinline__ void pl_store(u32* __restrict__ g, size_t idx, const Fe<P>& v) {   // v canonical
  u32 w[P::NW];
  fe_pack<P>(v, w);
  uint4* p4 = reinterpret_cast<uint4*>(g + idx * P::NW);
#pragma unroll
  for (int q = 0; q < P::NW / 4; q++) p4[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
// plain-domain helpers: canonical in, canonical out
template <class P> __device__ __forceinline__ Fe<P> pl_mul(const Fe<P>& x, const Fe<P>& y) { return fe_reduce<P>(FeAsm<P>::mul(FeAsm<P>::mul(x, y), fe_r2<P>())); }
template <class P> __device__ __forceinline__ Fe<P> pl_add(const Fe<P>& x, const Fe<P>& y) { return fe_reduce<P>(fe_carry<P>(fe_add<P>(x, y))); }
template <class P> __device__ __forceinline__ Fe<P> pl_sub(const Fe<P>& x, const Fe<P>& y) { return fe_reduce<P>(fe_carry<P>(fe_sub<P, 2>(x, y))); }
template <class P> __device__ __forceinline__ Fe<P> pl_one() { Fe<P> r = fe_zero<P>(); r.l[0] = 1; return r; }
template <class P> __device__ __forceinline__ Fe<P> shfl_fe(const Fe<P>& v, int src) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = (u32)__shfl((int)v.l[i], src);
  return r;
}
template <class P> __device__ __forceinline__ Fe<P> shfl_up1_fe(const Fe<P>& v) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = (u32)__shfl_up((int)v.l[i], 1);
  return r;
}
template <class P> __device__ __forceinline__ Fe<P> shfl_xor_fe(const Fe<P>& v, int m) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.l[i] = (u32)__shfl_xor((int)v.l[i], m);
  return r;
}

// ---- level 0: monic zerofier of 64 points per wave --------------------------------------------------------
// lane j holds coefficient j.  Multiplying by (X - d): c_j <- c_{j-1} - d c_j.  After 64 steps the (implicit) leading
// coefficient has left lane 63; low[chunk * 64 + j] = coefficient j.  Points beyond n are 0 (padding).
template <class P>
__global__ __launch_bounds__(64) void k_chunk_zerofier(const u32* __restrict__ domain, size_t n, u32* __restrict__ low) {
  const int lane = threadIdx.x;
  const size_t chunk = blockIdx.x;
  const size_t idx = chunk * CHUNK + lane;
  Fe<P> d = fe_zero<P>();
  if (idx < n) d = fe_reduce<P>(fe_to_mont<P>(pl_load<P>(domain, idx)));     // Montgomery form: fe_mul(c, dM) = c d in the plain domain
  Fe<P> c = fe_zero<P>();
  if (lane == 0) c = pl_one<P>();
  for (int i = 0; i < CHUNK; i++) {
    const Fe<P> di = shfl_fe<P>(d, i);
    Fe<P> up = shfl_up1_fe<P>(c);
    if (lane == 0) up = fe_zero<P>();
    c = pl_sub<P>(up, fe_reduce<P>(FeAsm<P>::mul(c, di)));
  }
  pl_store<P>(low, idx, c);
}

// ---- level step kernels -------------------------------------------------------------------------------------
// T[p * 2D + i] = i < D ? src[p * D + i] : 0
template <class P>
__global__ __launch_bounds__(256) void k_pad_double(const u32* __restrict__ src, int lgD, size_t total2, u32* __restrict__ T) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total2) return;
  const size_t p = e >> (lgD + 1), i = e & (((size_t)2 << lgD) - 1);
  Fe<P> v = fe_zero<P>();
  if (i < ((size_t)1 << lgD)) v = pl_load<P>(src, (p << lgD) + i);
  pl_store<P>(T, e, v);
}
// U[q * 2D + k] = T[(2q) * 2D + k] * T[(2q + 1) * 2D + k]
template <class P>
__global__ __launch_bounds__(256) void k_pair_mul(const u32* __restrict__ T, int lgD, size_t total, u32* __restrict__ U) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t q = e >> (lgD + 1), k = e & (((size_t)2 << lgD) - 1);
  const size_t a = ((2 * q) << (lgD + 1)) + k;
  pl_store<P>(U, e, pl_mul<P>(pl_load<P>(T, a), pl_load<P>(T, a + ((size_t)2 << lgD))));
}
// next[q * 2D + i] = U[q * 2D + i] + (i >= D ? cur[2q D + i - D] + cur[(2q + 1) D + i - D] : 0)
template <class P>
__global__ __launch_bounds__(256) void k_monic_fixup(const u32* __restrict__ U, const u32* __restrict__ cur, int lgD, size_t total, u32* __restrict__ next) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t D = (size_t)1 << lgD;
  const size_t q = e >> (lgD + 1), i = e & (2 * D - 1);
  Fe<P> v = pl_load<P>(U, e);
  if (i >= D) {
    const size_t a = ((2 * q) << lgD) + (i - D);
    v = pl_add<P>(v, pl_add<P>(pl_load<P>(cur, a), pl_load<P>(cur, a + D)));
  }
  pl_store<P>(next, e, v);
}

// ---- power-series inverse of g = rev(Z_pad) (g[0] = 1, g[k] = Zpad[N - k]) ----------------------------------------
// g as a coefficient array: gk[k], k < N (g[N] = Zpad[0] is never needed mod X^N)
template <class P>
__global__ __launch_bounds__(256) void k_reverse_monic(const u32* __restrict__ low_top, size_t N, u32* __restrict__ g) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  pl_store<P>(g, k, k == 0 ? pl_one<P>() : pl_load<P>(low_top, N - k));
}
// inv[0 .. 64) of 1 / g, one wave: inv_k = - sum_{j=1..k} g_j inv_{k-j}; lane j holds g_j, the partial sums are
// wave-reduced (64 steps)
template <class P>
__global__ __launch_bounds__(64) void k_series_inverse_64(const u32* __restrict__ g, u32* __restrict__ inv) {
  __shared__ u32 sh[CHUNK * P::NW];
  const int lane = threadIdx.x;
  const Fe<P> gj = pl_load<P>(g, lane);
  if (lane == 0) pl_store<P>(sh, 0, pl_one<P>());
  __syncthreads();
  for (int k = 1; k < CHUNK; k++) {
    Fe<P> term = fe_zero<P>();
    if (lane >= 1 && lane <= k) term = pl_mul<P>(gj, pl_load<P>(sh, k - lane));
    for (int m = 1; m < 64; m <<= 1) term = pl_add<P>(term, shfl_xor_fe<P>(term, m));
    if (lane == 0) pl_store<P>(sh, k, fe_reduce<P>(fe_neg_canon<P>(term)));
    __syncthreads();
  }
  pl_store<P>(inv, lane, pl_load<P>(sh, lane));
}
// A[i] = i < m_in ? src[i] : 0 for i < 2m   (src may be longer than m_in)
template <class P>
__global__ __launch_bounds__(256) void k_copy_pad(const u32* __restrict__ src, size_t m_in, size_t total, u32* __restrict__ dst) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  pl_store<P>(dst, e, e < m_in ? pl_load<P>(src, e) : fe_zero<P>());
}
// dst[i] = a[i] * b[i]
template <class P>
__global__ __launch_bounds__(256) void k_mul_vec(const u32* __restrict__ a, const u32* __restrict__ b, size_t total, u32* __restrict__ dst) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  pl_store<P>(dst, e, pl_mul<P>(pl_load<P>(a, e), pl_load<P>(b, e)));
}
// dst[i] = i < m ? src[m + i] : 0 for i < 2m  (the error term e of g inv = 1 + X^m e)
template <class P>
__global__ __launch_bounds__(256) void k_upper_half_pad(const u32* __restrict__ src, size_t m, u32* __restrict__ dst) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * m) return;
  pl_store<P>(dst, e, e < m ? pl_load<P>(src, m + e) : fe_zero<P>());
}
// inv[m + i] = - prod[i], i < m
template <class P>
__global__ __launch_bounds__(256) void k_newton_store(const u32* __restrict__ prod, size_t m, u32* __restrict__ inv) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  pl_store<P>(inv, m + e, fe_reduce<P>(fe_neg_canon<P>(pl_load<P>(prod, e))));
}

// ---- transposed evaluation ------------------------------------------------------------------------------------
// A[i] = f_rev[i] = f[N - 1 - i] (f has m <= N coefficients, zero beyond), B[i] = inv[i], both zero on [N, 2N)
template <class P>
__global__ __launch_bounds__(256) void k_eval_prepare(const u32* __restrict__ f, size_t m, const u32* __restrict__ inv, size_t N, u32* __restrict__ A,
                                                      u32* __restrict__ B) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * N) return;
  Fe<P> a = fe_zero<P>(), b = fe_zero<P>();
  if (e < N) {
    const size_t src = N - 1 - e;
    if (src < m) a = pl_load<P>(f, src);
    b = pl_load<P>(inv, e);
  }
  pl_store<P>(A, e, a);
  pl_store<P>(B, e, b);
}
// t[i] = prod[N - 1 - i], i < N
template <class P>
__global__ __launch_bounds__(256) void k_reverse_low(const u32* __restrict__ prod, size_t N, u32* __restrict__ t) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  pl_store<P>(t, e, pl_load<P>(prod, N - 1 - e));
}
// down step, children of degree D: V[c * 2D + k] = That_node[(c >> 1) * 2D + k] * (Zhat[(c ^ 1) * 2D + k] + (-1)^k)
template <class P>
__global__ __launch_bounds__(256) void k_down_mul(const u32* __restrict__ That, const u32* __restrict__ Zhat, int lgD, size_t total2, u32* __restrict__ V) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total2) return;
  const size_t c = e >> (lgD + 1), k = e & (((size_t)2 << lgD) - 1);
  const Fe<P> t = pl_load<P>(That, ((c >> 1) << (lgD + 1)) + k);
  Fe<P> z = pl_load<P>(Zhat, ((c ^ 1) << (lgD + 1)) + k);
  z = (k & 1) ? pl_sub<P>(z, pl_one<P>()) : pl_add<P>(z, pl_one<P>());       // + NTT(X^D)[k] = w^(D k) = (-1)^k
  pl_store<P>(V, e, pl_mul<P>(t, z));
}
// t_child[c * D + i] = V[c * 2D + D + i]
template <class P>
__global__ __launch_bounds__(256) void k_take_upper(const u32* __restrict__ V, int lgD, size_t total, u32* __restrict__ t) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t c = e >> lgD, i = e & (((size_t)1 << lgD) - 1);
  pl_store<P>(t, e, pl_load<P>(V, (c << (lgD + 1)) + ((size_t)1 << lgD) + i));
}
// 64-point finish: f(x_i) = sum_k t[k] q_k, q_0 = 1, q_k = g_k + x_i q_{k-1}, g_k = Zc[64 - k] (g_64 unused)
template <class P>
__global__ __launch_bounds__(64) void k_chunk_eval(const u32* __restrict__ domain, size_t n, const u32* __restrict__ low0, const u32* __restrict__ t,
                                                   u32* __restrict__ out, size_t out_n) {
  __shared__ u32 sh_t[CHUNK * P::NW], sh_z[CHUNK * P::NW];
  const int lane = threadIdx.x;
  const size_t idx = (size_t)blockIdx.x * CHUNK + lane;
  pl_store<P>(sh_t, lane, pl_load<P>(t, idx));
  pl_store<P>(sh_z, lane, pl_load<P>(low0, idx));
  __syncthreads();
  Fe<P> x = fe_zero<P>();
  if (idx < n) x = fe_reduce<P>(fe_to_mont<P>(pl_load<P>(domain, idx)));
  Fe<P> q = pl_one<P>();
  Fe<P> acc = pl_load<P>(sh_t, 0);
  for (int k = 1; k < CHUNK; k++) {
    q = pl_add<P>(pl_load<P>(sh_z, CHUNK - k), fe_reduce<P>(FeAsm<P>::mul(q, x)));
    acc = pl_add<P>(acc, pl_mul<P>(pl_load<P>(sh_t, k), q));
  }
  if (idx < out_n) pl_store<P>(out, idx, acc);
}

// ---- interpolation ------------------------------------------------------------------------------------------------
// dz[j] = (j + 1) * Zreal[j + 1], Zreal[j] = Zpad[j + pad] (Zpad[N] = 1), j < n
template <class P>
__global__ __launch_bounds__(256) void k_derivative(const u32* __restrict__ low_top, size_t N, size_t pad, size_t n, u32* __restrict__ dz) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const size_t src = j + 1 + pad;
  const Fe<P> z = src < N ? pl_load<P>(low_top, src) : pl_one<P>();
  Fe<P> m = fe_zero<P>();
  const u64 jj = (u64)j + 1;
  m.l[0] = (u32)(jj & MASK29); m.l[1] = (u32)((jj >> 29) & MASK29); m.l[2] = (u32)(jj >> 58);
  pl_store<P>(dz, j, pl_mul<P>(z, m));
}
// P_chunk = sum_i w_i Zc / (X - x_i): lane i divides synthetically (q_63 = 1, q_{k-1} = zc_k + x_i q_k), the scaled
// coefficients are summed over the wave; out0[chunk * 64 + k] = coefficient k.
// The 64 sums over the wave go through LDS eight coefficients at a time (round 5): every lane writes its eight scaled terms, lane
// (col, sub) adds rows sub, sub + 8, ... of column col, three shuffle steps finish the column -- 8 LDS reads and 3 shuffles per eight
// coefficients and lane where the butterfly reduction of each coefficient took 6 shuffle steps of NW words each (k_chunk_combine was
// 0.35 ms of a 1.63-ms interpolation of 16 registers at 2^14 points).
constexpr int COMBINE_BLOCK = 8;
template <class P>
__global__ __launch_bounds__(64) void k_chunk_combine(const u32* __restrict__ domain, const u32* __restrict__ w, size_t n, const u32* __restrict__ low0,
                                                      u32* __restrict__ out0, size_t chunks) {
  // blockIdx.x = register * chunks + chunk: the weights and the output of register r are N = chunks * 64 elements apart,
  // the domain and the level-0 zerofiers are shared
  static_assert(CHUNK == 64 && COMBINE_BLOCK * COMBINE_BLOCK == CHUNK, "one wave, eight columns of eight row groups");
  __shared__ u32 sh_z[CHUNK * P::NW];
  __shared__ u32 sh_t[COMBINE_BLOCK * CHUNK * P::NW];      // [column][lane]: the terms of eight coefficients
  __shared__ u32 sh_o[CHUNK * P::NW];                      // the chunk's coefficients, for the coalesced store
  const int lane = threadIdx.x;
  const size_t reg = blockIdx.x / chunks;
  const size_t idx = ((size_t)blockIdx.x - reg * chunks) * CHUNK + lane;
  w += reg * chunks * CHUNK * P::NW;
  out0 += reg * chunks * CHUNK * P::NW;
  pl_store<P>(sh_z, lane, pl_load<P>(low0, idx));
  __syncthreads();
  Fe<P> x = fe_zero<P>(), wi = fe_zero<P>();
  if (idx < n) { x = fe_reduce<P>(fe_to_mont<P>(pl_load<P>(domain, idx))); wi = fe_reduce<P>(fe_to_mont<P>(pl_load<P>(w, idx))); }
  Fe<P> q = pl_one<P>();
  const int col = lane >> 3, sub = lane & 7;
  for (int kb = CHUNK - COMBINE_BLOCK; kb >= 0; kb -= COMBINE_BLOCK) {
#pragma unroll 1
    for (int kk = COMBINE_BLOCK - 1; kk >= 0; kk--) {
      const int k = kb + kk;
      pl_store<P>(sh_t, kk * CHUNK + lane, fe_reduce<P>(FeAsm<P>::mul(q, wi)));
      if (k > 0) q = pl_add<P>(pl_load<P>(sh_z, k), fe_reduce<P>(FeAsm<P>::mul(q, x)));
    }
    __syncthreads();
    Fe<P> sum = pl_load<P>(sh_t, col * CHUNK + sub);
#pragma unroll 1
    for (int j = 1; j < COMBINE_BLOCK; j++) sum = pl_add<P>(sum, pl_load<P>(sh_t, col * CHUNK + j * COMBINE_BLOCK + sub));
    for (int m = 1; m < COMBINE_BLOCK; m <<= 1) sum = pl_add<P>(sum, shfl_xor_fe<P>(sum, m));
    if (sub == 0) pl_store<P>(sh_o, kb + col, sum);
    __syncthreads();
  }
  pl_store<P>(out0, idx, pl_load<P>(sh_o, lane));
}
// W[q * 2D + k] = Pl[k] Zr[k] + Pr[k] Zl[k]   (all transformed, children q*2, q*2+1 of degree D)
template <class P>
__global__ __launch_bounds__(256) void k_up_mul(const u32* __restrict__ Phat, const u32* __restrict__ Zhat, int lgD, size_t total, u32* __restrict__ W,
                                                size_t zmask) {
  // several registers back to back share one tree: Phat / W run over all of them, Zhat (2N elements) is indexed modulo 2N
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t q = e >> (lgD + 1), k = e & (((size_t)2 << lgD) - 1);
  const size_t a = ((2 * q) << (lgD + 1)) + k, b = a + ((size_t)2 << lgD);
  pl_store<P>(W, e, pl_add<P>(pl_mul<P>(pl_load<P>(Phat, a), pl_load<P>(Zhat, b & zmask)), pl_mul<P>(pl_load<P>(Phat, b), pl_load<P>(Zhat, a & zmask))));
}

// ---- host orchestration ----------------------------------------------------------------------------------------
// Device scratch of the tree routines.  A call makes a dozen buffers whose sizes repeat from call to call; hipMalloc /
// hipFree cost more than the kernels of a small tree (and hipFree synchronises the device), so released blocks wait in a
// per-context pool and are handed out again (everything here runs on the context's own stream, so reuse is stream-ordered).
struct PolyPool {
  struct Block { void* p; size_t cap; };
  std::vector<Block> free_blocks;
  size_t bytes = 0;
};
static PolyPool g_poly_pool[MZK_MAX_CTX];
constexpr size_t POLY_POOL_MAX_BYTES = (size_t)2 << 30;
void poly_release_plans();
size_t poly_trim_idle();
static void poly_release_free_blocks() {
  PolyPool& pool = g_poly_pool[ctx().index];
  for (auto& b : pool.free_blocks) (void)hipFree(b.p);
  pool.free_blocks.clear();
  pool.bytes = 0;
}
void poly_release_pool() {         // shutdown / mzk_trim_workspace: nothing of this file is in use
  poly_release_plans();            // the cached interpolation plans: their buffers go back into the pool first
  poly_release_free_blocks();
}
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  ~DevBuf() {
    if (!p) return;
    PolyPool& pool = g_poly_pool[ctx().index];
    if (pool.bytes + cap > POLY_POOL_MAX_BYTES) { (void)hipFree(p); return; }
    pool.free_blocks.push_back({p, cap});
    pool.bytes += cap;
  }
  int alloc(size_t bytes) {
    size_t want = 4096;
    while (want < bytes) want <<= 1;                 // sizes are N * element size: powers of two anyway
    PolyPool& pool = g_poly_pool[ctx().index];
    for (size_t i = pool.free_blocks.size(); i-- > 0;) {
      if (pool.free_blocks[i].cap == want) {
        p = pool.free_blocks[i].p; cap = want;
        pool.bytes -= want;
        pool.free_blocks.erase(pool.free_blocks.begin() + (long)i);
        return MZK_OK;
      }
    }
    if (hipMalloc(&p, want) != hipSuccess) {
      // the ABI's out-of-memory contract (mzk.h): give back what the running call does not use -- the blocks parked in this pool, the
      // cached interpolation plans but the one in use, then (dev_alloc) the idle workspace slots -- retry, and what is still refused is
      // MZK_E_NOMEM, never a bare HIP error
      (void)hipGetLastError();
      p = nullptr;
      (void)poly_trim_idle();
      MZK_TRY(dev_alloc(&p, want, "polynomial scratch"));
    }
    cap = want;
    return MZK_OK;
  }
  u32* w() const { return (u32*)p; }
};
static inline unsigned grid256(size_t total) { return (unsigned)((total + 255) / 256); }

template <class P> struct PolyTree {
  int fid;
  size_t n, N, pad;
  int levels;                       // number of level steps above level 0: N = 64 << levels
  const HostField* hf;
  uint64_t root[4]; size_t root_order;
  hipStream_t s;
  std::vector<DevBuf> low;          // low[l]: N elements, polynomials of degree 64 << l
  std::vector<DevBuf> Zhat;         // Zhat[l] (l < levels): 2N elements, NTT_{2D}(low part) of every level-l polynomial
  DevBuf U;                         // N scratch

  void sub_root(size_t size, uint64_t* out) const {     // primitive `size`-th root from the caller's root
    uint64_t r[4] = {0, 0, 0, 0};
    memcpy(r, root, 8 * hf->nl);
    for (size_t o = root_order; o > size; o >>= 1) h_mulmod(hf, r, r, r);
    memcpy(out, r, 8 * hf->nl);
  }
  int ntt(void* buf, size_t size, size_t batch, int inverse) const {
    uint64_t r[4];
    sub_root(size, r);
    return ntt_batch_dev_impl(fid, r, buf, buf, size, batch, inverse, s);
  }
  // builds every level; keep_hats: retain the transformed low parts (evaluate / interpolate need them)
  int build(const void* d_domain, bool keep_hats) {
    const size_t esz = field_bytes(fid);
    low.resize(levels + 1);
    Zhat.resize(levels);
    MZK_TRY(low[0].alloc(N * esz));
    hipLaunchKernelGGL((k_chunk_zerofier<P>), dim3((unsigned)(N / CHUNK)), dim3(64), 0, s, (const u32*)d_domain, n, low[0].w());
    if (levels) MZK_TRY(U.alloc(N * esz));
    DevBuf shared_T;
    if (levels && !keep_hats) MZK_TRY(shared_T.alloc(2 * N * esz));
    for (int l = 0; l < levels; l++) {
      const int lgD = 6 + l;
      u32* T;
      if (keep_hats) { MZK_TRY(Zhat[l].alloc(2 * N * esz)); T = Zhat[l].w(); } else T = shared_T.w();
      MZK_TRY(low[l + 1].alloc(N * esz));
      hipLaunchKernelGGL((k_pad_double<P>), dim3(grid256(2 * N)), dim3(256), 0, s, (const u32*)low[l].w(), lgD, 2 * N, T);
      MZK_TRY(ntt(T, (size_t)2 << lgD, N >> lgD, 0));
      hipLaunchKernelGGL((k_pair_mul<P>), dim3(grid256(N)), dim3(256), 0, s, (const u32*)T, lgD, N, U.w());
      MZK_TRY(ntt(U.p, (size_t)2 << lgD, N >> (lgD + 1), 1));
      hipLaunchKernelGGL((k_monic_fixup<P>), dim3(grid256(N)), dim3(256), 0, s, (const u32*)U.w(), (const u32*)low[l].w(), lgD, N, low[l + 1].w());
    }
    MZK_HIP(hipGetLastError());
    if (!keep_hats) MZK_HIP(hipStreamSynchronize(s));              // shared_T is freed on return
    return MZK_OK;
  }
  // values of f (m <= N coefficients, device) at the N padded points -> d_out[0 .. out_n)
  int evaluate(const void* d_f, size_t m, const void* d_domain, void* d_out, size_t out_n) {
    const size_t esz = field_bytes(fid);
    DevBuf g, inv, A, B, t, V;
    MZK_TRY(g.alloc(N * esz)); MZK_TRY(inv.alloc(N * esz)); MZK_TRY(A.alloc(2 * N * esz)); MZK_TRY(B.alloc(2 * N * esz));
    MZK_TRY(t.alloc(N * esz)); MZK_TRY(V.alloc(2 * N * esz));
    // 1 / rev(Z_pad) mod X^N
    hipLaunchKernelGGL((k_reverse_monic<P>), dim3(grid256(N)), dim3(256), 0, s, (const u32*)low[levels].w(), N, g.w());
    hipLaunchKernelGGL((k_series_inverse_64<P>), dim3(1), dim3(64), 0, s, (const u32*)g.w(), inv.w());
    for (size_t mm = CHUNK; mm < N; mm <<= 1) {      // inv mod X^mm -> mod X^2mm
      hipLaunchKernelGGL((k_copy_pad<P>), dim3(grid256(2 * mm)), dim3(256), 0, s, (const u32*)g.w(), 2 * mm, 2 * mm, A.w());
      hipLaunchKernelGGL((k_copy_pad<P>), dim3(grid256(2 * mm)), dim3(256), 0, s, (const u32*)inv.w(), mm, 2 * mm, B.w());
      MZK_TRY(ntt(A.p, 2 * mm, 1, 0));
      MZK_TRY(ntt(B.p, 2 * mm, 1, 0));
      hipLaunchKernelGGL((k_mul_vec<P>), dim3(grid256(2 * mm)), dim3(256), 0, s, (const u32*)A.w(), (const u32*)B.w(), 2 * mm, A.w());
      MZK_TRY(ntt(A.p, 2 * mm, 1, 1));                                   // g inv_m mod (X^2m - 1): coefficients [m, 2m) are e
      hipLaunchKernelGGL((k_upper_half_pad<P>), dim3(grid256(2 * mm)), dim3(256), 0, s, (const u32*)A.w(), mm, V.w());
      MZK_TRY(ntt(V.p, 2 * mm, 1, 0));
      hipLaunchKernelGGL((k_mul_vec<P>), dim3(grid256(2 * mm)), dim3(256), 0, s, (const u32*)V.w(), (const u32*)B.w(), 2 * mm, V.w());
      MZK_TRY(ntt(V.p, 2 * mm, 1, 1));                                   // e inv_m: low m coefficients exact
      hipLaunchKernelGGL((k_newton_store<P>), dim3(grid256(mm)), dim3(256), 0, s, (const u32*)V.w(), mm, inv.w());
    }
    // t_root[i] = (rev(f) inv)[N - 1 - i]
    hipLaunchKernelGGL((k_eval_prepare<P>), dim3(grid256(2 * N)), dim3(256), 0, s, (const u32*)d_f, m, (const u32*)inv.w(), N, A.w(), B.w());
    MZK_TRY(ntt(A.p, 2 * N, 1, 0));
    MZK_TRY(ntt(B.p, 2 * N, 1, 0));
    hipLaunchKernelGGL((k_mul_vec<P>), dim3(grid256(2 * N)), dim3(256), 0, s, (const u32*)A.w(), (const u32*)B.w(), 2 * N, A.w());
    MZK_TRY(ntt(A.p, 2 * N, 1, 1));
    hipLaunchKernelGGL((k_reverse_low<P>), dim3(grid256(N)), dim3(256), 0, s, (const u32*)A.w(), N, t.w());
    // down the tree: node degree 2D -> children degree D
    for (int l = levels - 1; l >= 0; l--) {
      const int lgD = 6 + l;
      MZK_TRY(ntt(t.p, (size_t)2 << lgD, N >> (lgD + 1), 0));
      hipLaunchKernelGGL((k_down_mul<P>), dim3(grid256(2 * N)), dim3(256), 0, s, (const u32*)t.w(), (const u32*)Zhat[l].w(), lgD, 2 * N, V.w());
      MZK_TRY(ntt(V.p, (size_t)2 << lgD, N >> lgD, 1));
      hipLaunchKernelGGL((k_take_upper<P>), dim3(grid256(N)), dim3(256), 0, s, (const u32*)V.w(), lgD, N, t.w());
    }
    hipLaunchKernelGGL((k_chunk_eval<P>), dim3((unsigned)(N / CHUNK)), dim3(64), 0, s, (const u32*)d_domain, n, (const u32*)low[0].w(), (const u32*)t.w(),
                       (u32*)d_out, out_n);
    MZK_HIP(hipGetLastError());
    MZK_HIP(hipStreamSynchronize(s));      // the temporaries above are freed on return
    return MZK_OK;
  }
  // sum_i w_i Z_pad / (X - x_i) -> N coefficients in d_out (weights of padded points are ignored: treated as 0)
  // d_w: regs weight vectors, N elements apart; d_out: regs results of N coefficients each
  int combine(const void* d_domain, const void* d_w, void* d_out, size_t regs = 1) {
    const size_t esz = field_bytes(fid);
    const size_t M = N * regs;
    DevBuf cur, nxt, Ph, W;
    MZK_TRY(cur.alloc(M * esz)); MZK_TRY(nxt.alloc(M * esz)); MZK_TRY(Ph.alloc(2 * M * esz)); MZK_TRY(W.alloc(M * esz));
    hipLaunchKernelGGL((k_chunk_combine<P>), dim3((unsigned)(M / CHUNK)), dim3(64), 0, s, (const u32*)d_domain, (const u32*)d_w, n, (const u32*)low[0].w(), cur.w(),
                       N / CHUNK);
    for (int l = 0; l < levels; l++) {
      const int lgD = 6 + l;
      hipLaunchKernelGGL((k_pad_double<P>), dim3(grid256(2 * M)), dim3(256), 0, s, (const u32*)cur.w(), lgD, 2 * M, Ph.w());
      MZK_TRY(ntt(Ph.p, (size_t)2 << lgD, M >> lgD, 0));
      hipLaunchKernelGGL((k_up_mul<P>), dim3(grid256(M)), dim3(256), 0, s, (const u32*)Ph.w(), (const u32*)Zhat[l].w(), lgD, M, W.w(), 2 * N - 1);
      MZK_TRY(ntt(W.p, (size_t)2 << lgD, M >> (lgD + 1), 1));
      hipLaunchKernelGGL((k_monic_fixup<P>), dim3(grid256(M)), dim3(256), 0, s, (const u32*)W.w(), (const u32*)cur.w(), lgD, M, nxt.w());
      std::swap(cur.p, nxt.p);
    }
    MZK_HIP(hipMemcpyAsync(d_out, cur.p, M * esz, hipMemcpyDeviceToDevice, s));
    MZK_HIP(hipGetLastError());
    MZK_HIP(hipStreamSynchronize(s));
    return MZK_OK;
  }
};

static size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }

// shared argument checks: the reference's two assertions (ntt.rs:122-123 etc.) and canonical inputs
static int check_root(const HostField* hf, const uint64_t* root, size_t root_order) {
  if (!h_is_canonical(hf, root)) { set_error("poly: root not canonical"); return MZK_E_RANGE; }
  uint64_t t[4];
  h_powmod_u64(hf, t, root, root_order);
  if (!h_is_one(hf, t)) { set_error("assertion failed: primitive_root.pow(root_order).is_one()"); return MZK_E_ROOT_ORDER; }
  h_powmod_u64(hf, t, root, root_order / 2);
  if (h_is_one(hf, t)) { set_error("assertion failed: !primitive_root.pow(root_order / 2).is_one()"); return MZK_E_ROOT_PRIM; }
  return MZK_OK;
}
static int check_canonical(const HostField* hf, const uint64_t* v, size_t n, const char* what) {
  for (size_t i = 0; i < n; i++) if (!h_is_canonical(hf, v + i * hf->nl)) { set_error("poly: %s[%zu] not canonical", what, i); return MZK_E_RANGE; }
  return MZK_OK;
}
template <class P> static int tree_init(PolyTree<P>* T, int fid, size_t n, const uint64_t* root, size_t root_order, hipStream_t s) {
  T->fid = fid; T->n = n; T->hf = host_field(fid); T->s = s;
  T->N = next_pow2(n < (size_t)CHUNK ? (size_t)CHUNK : n);
  T->pad = T->N - n;
  T->levels = 0;
  while (((size_t)CHUNK << T->levels) < T->N) T->levels++;
  // The results do not depend on which root drives the internal transforms (only the reference's assertions and its
  // output lengths look at the caller's): use the field's own 2^28-th root so that the 2N-point products of the
  // transposed evaluation never run out of order.
  (void)root; (void)root_order;
  const unsigned lg = 28;
  memset(T->root, 0, sizeof T->root);
  MZK_TRY(mzk_root_of_unity(fid, lg, T->root));
  T->root_order = (size_t)1 << lg;
  if (2 * T->N > T->root_order) { set_error("poly: %zu points exceed the 2^27 supported by the internal transforms", n); return MZK_E_ARG; }
  return MZK_OK;
}
// the reference's own requirement on root_order: its zerofiers of k >= 8 points go through fast_multiply's NTT path with
// order next_pow2(k + 1) <= root_order, otherwise the product wraps around or the inner ntt panics (ntt.rs:90-105)
static int check_order_for(size_t points, size_t root_order, const char* who) {
  if (points < 8) return MZK_OK;
  size_t ord = 1;
  while (ord < points + 1) ord <<= 1;
  if (root_order < ord) {
    set_error("%s: a zerofier of %zu points needs a root of order >= %zu (got %zu): the reference's product wraps around or its inner ntt panics", who, points, ord, root_order);
    return MZK_E_LENGTH;
  }
  return MZK_OK;
}

// the zerofier of a subgroup prefix without the tree (defined with the other prefix routines below); *done = false: not such a domain
template <class P> static int zerofier_of_prefix(int fid, const uint64_t* domain, size_t n, size_t len, uint64_t* out, hipStream_t s, bool* done);

template <class P>
static int zerofier_impl(int fid, const uint64_t* domain, size_t n, const uint64_t* root, size_t root_order, uint64_t* out, size_t* out_len) {
  const HostField* hf = host_field(fid);
  const int nl = hf->nl;
  const size_t esz = field_bytes(fid);
  if (n == 0) { *out_len = 0; return MZK_OK; }                              // ntt.rs:125-127
  const size_t ord = next_pow2(n + 1);
  const size_t len = n < 8 ? n + 1 : ord;                                   // degree < 8: `lhs * rhs`, trimmed (ntt.rs:86-88); else fast_multiply's untrimmed order
  MZK_TRY(check_order_for(n, root_order, "fast_zerofier"));
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  {
    bool done = false;
    MZK_TRY(zerofier_of_prefix<P>(fid, domain, n, len, out, s, &done));
    if (done) { *out_len = len; return MZK_OK; }
  }
  PolyTree<P> T;
  MZK_TRY(tree_init(&T, fid, n, root, root_order, s));
  DevBuf d_dom;
  MZK_TRY(d_dom.alloc(n * esz));
  MZK_HIP(hipMemcpyAsync(d_dom.p, domain, n * esz, hipMemcpyHostToDevice, s));
  MZK_TRY(T.build(d_dom.p, false));
  std::vector<uint64_t> top(T.N * nl);
  MZK_TRY(d2h_sync(top.data(), T.low[T.levels].p, T.N * esz, s));
  memset(out, 0, len * esz);
  for (size_t j = 0; j < n; j++) memcpy(out + j * nl, top.data() + (j + T.pad) * nl, esz);     // Z[j] = Zpad[j + pad]
  out[n * nl] = 1;                                                                             // monic
  *out_len = len;
  return MZK_OK;
}

template <class P>
static int evaluate_impl(int fid, const uint64_t* coef, size_t m, const uint64_t* domain, size_t n, const uint64_t* root, size_t root_order, uint64_t* out) {
  const HostField* hf = host_field(fid);
  const int nl = hf->nl;
  const size_t esz = field_bytes(fid);
  if (n == 0) return MZK_OK;
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  PolyTree<P> T;
  MZK_TRY(check_order_for(n - n / 2, root_order, "fast_evaluate"));           // the larger half's zerofier (ntt.rs:167-168)
  MZK_TRY(tree_init(&T, fid, n, root, root_order, s));
  DevBuf d_dom, d_f, d_vals;
  MZK_TRY(d_dom.alloc(n * esz)); MZK_TRY(d_f.alloc((m ? m : 1) * esz)); MZK_TRY(d_vals.alloc(T.N * esz));
  MZK_HIP(hipMemcpyAsync(d_dom.p, domain, n * esz, hipMemcpyHostToDevice, s));
  if (m) MZK_HIP(hipMemcpyAsync(d_f.p, coef, m * esz, hipMemcpyHostToDevice, s));
  MZK_TRY(T.build(d_dom.p, true));
  if (m <= T.N) {
    MZK_TRY(T.evaluate(d_f.p, m, d_dom.p, d_vals.p, n));
    MZK_TRY(d2h_sync(out, d_vals.p, n * esz, s));
    return MZK_OK;
  }
  // deg f >= N (more coefficients than padded points; no reference caller does this): f = sum_b X^(b N) f_b, so
  // f(x) = sum_b x^(b N) f_b(x) -- every block goes through the tree, Horner over the blocks on the host.
  std::vector<uint64_t> xn(n * nl), acc(n * nl, 0), blk(n * nl);
  for (size_t i = 0; i < n; i++) h_powmod_u64(hf, xn.data() + i * nl, domain + i * nl, (uint64_t)T.N);
  const size_t nb = (m + T.N - 1) / T.N;
  for (size_t b = nb; b-- > 0;) {
    const size_t lo = b * T.N, cnt = (m - lo < T.N) ? m - lo : T.N;
    MZK_TRY(T.evaluate((const uint8_t*)d_f.p + lo * esz, cnt, d_dom.p, d_vals.p, n));
    MZK_TRY(d2h_sync(blk.data(), d_vals.p, n * esz, s));
    for (size_t i = 0; i < n; i++) {       // acc = acc * x^N + f_b(x)   (n host products per block: parameter-sized next to the tree work)
      uint64_t t[4];
      h_mulmod(hf, t, acc.data() + i * nl, xn.data() + i * nl);
      // t + blk mod p
      unsigned __int128 c = 0;
      uint64_t r[4] = {0, 0, 0, 0};
      for (int k = 0; k < nl; k++) { c += (unsigned __int128)t[k] + blk[i * nl + k]; r[k] = (uint64_t)c; c >>= 64; }
      bool ge = c != 0;
      if (!ge) { ge = true; for (int k = nl - 1; k >= 0; k--) if (r[k] != hf->p[k]) { ge = r[k] > hf->p[k]; break; } }
      if (ge) { unsigned __int128 br = 0; for (int k = 0; k < nl; k++) { unsigned __int128 d = (unsigned __int128)r[k] - hf->p[k] - (uint64_t)br; r[k] = (uint64_t)d; br = (d >> 64) & 1; } }
      memcpy(acc.data() + i * nl, r, 8 * nl);
    }
  }
  memcpy(out, acc.data(), n * esz);
  return MZK_OK;
}

// `batch` value vectors over ONE domain (the registers of a trace, fast_stark.rs:203-215): the subproduct tree and the
// Z'(d_i) are built once, each register costs a pointwise division and one up-sweep.  out: batch rows of n elements
// (row r holds out_lens[r] coefficients, zero-padded).
// trimmed length of every row (the reference's final `+` trims trailing zeros, polynomial.rs:214-228): lens[r] = 1 + the highest
// i < n with rows[r * stride + pad + i] != 0, or 0.  One workgroup per row.
template <class P>
__global__ __launch_bounds__(256) void k_row_len(const u32* __restrict__ rows, size_t stride, size_t pad, size_t n, u32* __restrict__ lens) {
  __shared__ u32 sh[256];
  const u32* row = rows + ((size_t)blockIdx.x * stride + pad) * P::NW;
  u32 best = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) {
    u32 a = 0;
#pragma unroll
    for (int k = 0; k < P::NW; k++) a |= row[i * P::NW + k];
    if (a) best = (u32)i + 1;
  }
  sh[threadIdx.x] = best;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off && sh[threadIdx.x + off] > sh[threadIdx.x]) sh[threadIdx.x] = sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) lens[blockIdx.x] = sh[0];
}

// ---- interpolation over a prefix of a power-of-two subgroup ---------------------------------------------------------------------
// FastStark::prove interpolates every trace register over trace_domain = [omicron^i, i < cycles] (fast_stark.rs:197-215): the first n
// points of the subgroup generated by g = omicron, whose order N = 2^k is the next power of two above the cycle count (N - n = m points
// are missing: 3 for the reference's padded traces).  On the WHOLE subgroup interpolation is the inverse transform with root g; on a
// prefix the interpolant f (degree < n = N - m, unique: the points are distinct) is the inverse transform of the values extended by the
// m numbers u_j = f(g^(n+j)), and those are fixed by the m top coefficients of the transform being zero:
//     sum_j u_j x_(n+j)^(1+t) = - sum_(i<n) v_i x_i^(1+t),  t < m,   x_i = g^i      (coefficient N-1-t of the inverse transform, times N)
// an m x m system whose matrix A[t][j] = x_(n+j)^(1+t) depends on the domain alone.  With its inverse, u_j = sum_i v_i C[j][i] where
// C[j][i] = - sum_t Ainv[j][t] x_i^(1+t) is part of the plan: a call costs m dot products of length n per register and ONE batched
// inverse transform of N points -- O(n log n + m n) where the subproduct tree is O(n log^2 n) with ~50 launches.  The coefficients are
// those of the same polynomial the reference's recursion returns (ntt.rs:203-252), trimmed the same way; any other domain (and a prefix
// with more than PREFIX_MAX_MISSING points missing) goes through the tree below.
constexpr size_t PREFIX_MAX_MISSING = 64;
// C[j * n + i] = Montgomery form of  - x_i sum_t ainv[j * m + t] x_i^t  (Horner in x_i); blockIdx.y = j
template <class P>
__global__ __launch_bounds__(256) void k_prefix_weights(const u32* __restrict__ domain, size_t n, const u32* __restrict__ ainv, int m, u32* __restrict__ C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j = blockIdx.y;
  if (i >= n) return;
  const Fe<P> x = fe_reduce<P>(fe_to_mont<P>(pl_load<P>(domain, i)));       // Montgomery form: mul(acc, x) is the plain product
  Fe<P> acc = fe_zero<P>();
  for (int t = m - 1; t >= 0; t--) acc = pl_add<P>(fe_reduce<P>(FeAsm<P>::mul(acc, x)), pl_load<P>(ainv, j * (size_t)m + t));
  acc = fe_reduce<P>(fe_neg_canon<P>(fe_reduce<P>(FeAsm<P>::mul(acc, x))));
  pl_store<P>(C, j * n + i, fe_reduce<P>(fe_to_mont<P>(acc)));
}
// ext[r * N + i] = i < n ? v[r * n + i] : 0, N = 2^lgN  (the m missing values are written by k_prefix_missing afterwards)
template <class P>
__global__ __launch_bounds__(256) void k_prefix_pad(const u32* __restrict__ v, size_t n, int lgN, size_t total, u32* __restrict__ ext) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t r = e >> lgN, i = e & (((size_t)1 << lgN) - 1);
  pl_store<P>(ext, e, i < n ? pl_load<P>(v, r * n + i) : fe_zero<P>());
}
// ext[r * N + n + j] = sum_i v[r * n + i] C[j * n + i]: one workgroup per (missing point j = blockIdx.x, register r = blockIdx.y)
constexpr int PREFIX_NT = 1024;
template <class P>
__global__ __launch_bounds__(PREFIX_NT) void k_prefix_missing(const u32* __restrict__ v, size_t n, size_t N, const u32* __restrict__ C, u32* __restrict__ ext) {
  __shared__ u32 sh[(PREFIX_NT / 64) * P::NW];
  const size_t j = blockIdx.x, r = blockIdx.y;
  const u32* row = v + r * n * P::NW;
  const u32* cj = C + j * n * P::NW;
  Fe<P> acc = fe_zero<P>();
  for (size_t i = threadIdx.x; i < n; i += PREFIX_NT) acc = pl_add<P>(acc, fe_reduce<P>(FeAsm<P>::mul(pl_load<P>(row, i), pl_load<P>(cj, i))));
  for (int k = 1; k < 64; k <<= 1) acc = pl_add<P>(acc, shfl_xor_fe<P>(acc, k));
  if ((threadIdx.x & 63) == 0) pl_store<P>(sh, threadIdx.x >> 6, acc);
  __syncthreads();
  if (threadIdx.x < 64) {
    Fe<P> t = threadIdx.x < PREFIX_NT / 64 ? pl_load<P>(sh, threadIdx.x) : fe_zero<P>();
    for (int k = 1; k < PREFIX_NT / 64; k <<= 1) t = pl_add<P>(t, shfl_xor_fe<P>(t, k));
    if (threadIdx.x == 0) pl_store<P>(ext, r * N + n + j, t);
  }
}

// host side (parameters only): a + b, a - b mod p for canonical a, b
static void hp_add(const HostField* f, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t t[4] = {0, 0, 0, 0};
  unsigned __int128 c = 0;
  for (int i = 0; i < f->nl; i++) { c += (unsigned __int128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; }
  bool ge = c != 0;
  if (!ge) { ge = true; for (int k = f->nl - 1; k >= 0; k--) if (t[k] != f->p[k]) { ge = t[k] > f->p[k]; break; } }
  if (ge) { unsigned __int128 br = 0; for (int k = 0; k < f->nl; k++) { unsigned __int128 d = (unsigned __int128)t[k] - f->p[k] - (uint64_t)br; t[k] = (uint64_t)d; br = (d >> 64) & 1; } }
  for (int i = 0; i < f->nl; i++) r[i] = t[i];
}
static void hp_sub(const HostField* f, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t nb[4] = {0, 0, 0, 0};
  bool zero = true;
  for (int i = 0; i < f->nl; i++) zero = zero && b[i] == 0;
  if (!zero) { unsigned __int128 br = 0; for (int k = 0; k < f->nl; k++) { unsigned __int128 d = (unsigned __int128)f->p[k] - b[k] - (uint64_t)br; nb[k] = (uint64_t)d; br = (d >> 64) & 1; } }
  hp_add(f, r, a, nb);
}
// domain[i] == g^i for i < n with g = domain[1] of order exactly N = next_pow2(n), at most PREFIX_MAX_MISSING points of the subgroup
// missing.  Two steps: the host looks at the order of g and at three points (an arbitrary domain leaves here after a few short
// exponentiations), then a kernel compares EVERY point with its power of g (n host products were 0.3 ms at 2^14 points -- more than the
// zerofier below takes) and raises a flag on the first difference.
static bool prefix_candidate(const HostField* hf, const uint64_t* domain, size_t n, size_t* N_out) {
  static const int enabled = tune_int("MZK_INTERP_PREFIX", 1);      // tuning build: 0 = every domain through the tree (A/B, tests of the tree)
  if (!enabled || n < 2) return false;
  const int nl = hf->nl;
  const size_t N = next_pow2(n);
  if (N - n > PREFIX_MAX_MISSING || (N - n) * n * (size_t)nl * 8 > ((size_t)1 << 30)) return false;
  if (!h_is_one(hf, domain)) return false;
  const uint64_t* g = domain + nl;
  uint64_t t[4] = {0, 0, 0, 0};
  h_powmod_u64(hf, t, g, N);
  if (!h_is_one(hf, t)) return false;
  h_powmod_u64(hf, t, g, N / 2);
  if (h_is_one(hf, t)) return false;
  if (n > 2) {
    h_mulmod(hf, t, g, g);
    if (memcmp(t, domain + 2 * nl, 8 * (size_t)nl)) return false;
    h_powmod_u64(hf, t, g, (uint64_t)(n - 1));
    if (memcmp(t, domain + (n - 1) * nl, 8 * (size_t)nl)) return false;
  }
  *N_out = N;
  return true;
}
template <class P>
__global__ __launch_bounds__(256) void k_prefix_check(const u32* __restrict__ domain, size_t n, u32* __restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe<P> g = fe_reduce<P>(fe_to_mont<P>(pl_load<P>(domain, 1)));
  Fe<P> one = fe_zero<P>();
  one.l[0] = 1;
  const Fe<P> want = fe_reduce<P>(FeAsm<P>::mul(fe_reduce<P>(fe_pow_u64<P>(g, (u64)i)), one));      // g^i, plain and canonical
  const Fe<P> have = pl_load<P>(domain, i);
  u32 diff = 0;
#pragma unroll
  for (int k = 0; k < P::L; k++) diff |= want.l[k] ^ have.l[k];
  if (diff) atomicOr(flag, 1u);
}
// launches the comparison of the n device-resident points; *d_flag (one word, zeroed here) is non-zero afterwards if the domain is no prefix
template <class P> static int prefix_check_launch(const void* d_domain, size_t n, u32* d_flag, hipStream_t s) {
  MZK_HIP(hipMemsetAsync(d_flag, 0, 4, s));
  hipLaunchKernelGGL((k_prefix_check<P>), dim3(grid256(n)), dim3(256), 0, s, (const u32*)d_domain, n, d_flag);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
// Ainv (m x m, row-major, canonical) of A[t][j] = x_(n+j)^(1+t) by Gauss-Jordan on the host; false if a pivot is missing (cannot
// happen for distinct non-zero x: every leading minor is a Vandermonde determinant times a product of x's)
static bool prefix_system_inverse(const HostField* hf, const uint64_t* g, size_t n, size_t m, std::vector<uint64_t>* out) {
  const size_t nl = (size_t)hf->nl;
  std::vector<uint64_t> a(m * 2 * m * 4, 0);                  // [A | I], four limbs per entry
  auto at = [&](size_t row, size_t col) { return a.data() + (row * 2 * m + col) * 4; };
  for (size_t j = 0; j < m; j++) {
    uint64_t x[4] = {0, 0, 0, 0}, pw[4] = {0, 0, 0, 0};
    h_powmod_u64(hf, x, g, (uint64_t)(n + j));
    memcpy(pw, x, 8 * nl);
    for (size_t t = 0; t < m; t++) { memcpy(at(t, j), pw, 8 * nl); h_mulmod(hf, pw, pw, x); }
    at(j, m + j)[0] = 1;
  }
  auto is_zero = [&](const uint64_t* v) { for (size_t k = 0; k < nl; k++) if (v[k]) return false; return true; };
  for (size_t col = 0; col < m; col++) {
    size_t piv = col;
    while (piv < m && is_zero(at(piv, col))) piv++;
    if (piv == m) return false;
    if (piv != col) for (size_t k = 0; k < 2 * m * 4; k++) std::swap(a[col * 2 * m * 4 + k], a[piv * 2 * m * 4 + k]);
    uint64_t inv[4] = {0, 0, 0, 0};
    h_invmod(hf, inv, at(col, col));
    for (size_t k = 0; k < 2 * m; k++) h_mulmod(hf, at(col, k), at(col, k), inv);
    for (size_t row = 0; row < m; row++) {
      if (row == col || is_zero(at(row, col))) continue;
      uint64_t f[4] = {0, 0, 0, 0}, prod[4] = {0, 0, 0, 0};
      memcpy(f, at(row, col), 8 * nl);
      for (size_t k = 0; k < 2 * m; k++) { h_mulmod(hf, prod, at(col, k), f); hp_sub(hf, at(row, k), at(row, k), prod); }
    }
  }
  out->assign(m * m * nl, 0);
  for (size_t j = 0; j < m; j++) for (size_t t = 0; t < m; t++) memcpy(out->data() + (j * m + t) * nl, at(j, m + t), 8 * nl);
  return true;
}

// The zerofier of such a prefix (FastStark's transition zerofier, fast_stark.rs:53-57: omicron^i for i < cycles - 1): Z D = X^N - 1 with
// D = prod_j (X - x_(n+j)) over the m missing points, so the coefficients of Z are minus those of the power series 1 / D, and by partial
// fractions  z_k = sum_j rho_j w_j^(k+1),  k <= n,  w_j = 1 / x_(n+j),  rho_j = 1 / prod_(l != j) (x_(n+j) - x_(n+l))  (host, m^2 products).
// One launch: a lane owns ZERO_CHUNK consecutive coefficients, one exponentiation per missing point and lane.  params: m pairs (rho_j, w_j).
constexpr int ZERO_CHUNK = 8;
template <class P>
__global__ __launch_bounds__(256) void k_prefix_zerofier(const u32* __restrict__ params, int m, size_t count, u32* __restrict__ out) {
  const size_t k0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * ZERO_CHUNK;
  if (k0 >= count) return;
  Fe<P> acc[ZERO_CHUNK];
#pragma unroll
  for (int c = 0; c < ZERO_CHUNK; c++) acc[c] = fe_zero<P>();
  for (int j = 0; j < m; j++) {
    const Fe<P> w = fe_reduce<P>(fe_to_mont<P>(pl_load<P>(params, 2 * (size_t)j + 1)));         // Montgomery form: mul(t, w) is the plain product
    Fe<P> t = fe_reduce<P>(FeAsm<P>::mul(pl_load<P>(params, 2 * (size_t)j), fe_reduce<P>(fe_pow_u64<P>(w, (u64)k0 + 1))));      // rho_j w_j^(k0+1), plain
#pragma unroll
    for (int c = 0; c < ZERO_CHUNK; c++) {
      acc[c] = pl_add<P>(acc[c], t);
      t = fe_reduce<P>(FeAsm<P>::mul(t, w));
    }
  }
#pragma unroll
  for (int c = 0; c < ZERO_CHUNK; c++) if (k0 + c < count) pl_store<P>(out, k0 + c, acc[c]);
}
// (rho_j, w_j) for j < m, canonical, interleaved; false if two missing points coincide (cannot happen in a subgroup)
static bool prefix_zerofier_params(const HostField* hf, const uint64_t* g, size_t n, size_t N, std::vector<uint64_t>* out) {
  const size_t nl = (size_t)hf->nl, m = N - n;
  std::vector<uint64_t> x(m * 4, 0);
  for (size_t j = 0; j < m; j++) h_powmod_u64(hf, x.data() + 4 * j, g, (uint64_t)(n + j));
  out->assign(2 * m * nl, 0);
  for (size_t j = 0; j < m; j++) {
    uint64_t prod[4] = {1, 0, 0, 0}, d[4] = {0, 0, 0, 0}, inv[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0};
    for (size_t l = 0; l < m; l++) {
      if (l == j) continue;
      hp_sub(hf, d, x.data() + 4 * j, x.data() + 4 * l);
      bool zero = true;
      for (size_t k = 0; k < nl; k++) zero = zero && d[k] == 0;
      if (zero) return false;
      h_mulmod(hf, prod, prod, d);
    }
    h_invmod(hf, inv, prod);
    h_powmod_u64(hf, w, g, (uint64_t)(N - (n + j)));           // 1 / x_(n+j) = g^(N - n - j)
    memcpy(out->data() + (2 * j) * nl, inv, 8 * nl);
    memcpy(out->data() + (2 * j + 1) * nl, w, 8 * nl);
  }
  return true;
}

template <class P>
static int zerofier_of_prefix(int fid, const uint64_t* domain, size_t n, size_t len, uint64_t* out, hipStream_t s, bool* done) {
  const HostField* hf = host_field(fid);
  const size_t nl = (size_t)hf->nl, esz = field_bytes(fid);
  size_t N = 0;
  *done = false;
  if (!prefix_candidate(hf, domain, n, &N)) return MZK_OK;
  const size_t m = N - n;
  std::vector<uint64_t> params;
  if (m && !prefix_zerofier_params(hf, domain + nl, n, N, &params)) return MZK_OK;
  // every point is compared on the device; the coefficients are computed behind the comparison without waiting for its answer and
  // thrown away if it says no (one synchronize for both)
  DevBuf d_dom, d_flag, d_params, d_z;
  u32 flag = 1;
  MZK_TRY(d_dom.alloc(n * esz)); MZK_TRY(d_flag.alloc(4));
  std::vector<uint64_t> z;
  struct Settle {            // an error return below must not leave copies in flight that read or write this frame's vectors
    hipStream_t s; bool armed;
    ~Settle() { if (armed) (void)hipStreamSynchronize(s); }
  } settle{s, true};
  MZK_HIP(hipMemcpyAsync(d_dom.p, domain, n * esz, hipMemcpyHostToDevice, s));
  MZK_TRY(prefix_check_launch<P>(d_dom.p, n, (u32*)d_flag.p, s));
  if (m) {
    MZK_TRY(d_params.alloc(2 * m * esz)); MZK_TRY(d_z.alloc((n + 1) * esz));
    MZK_HIP(hipMemcpyAsync(d_params.p, params.data(), 2 * m * esz, hipMemcpyHostToDevice, s));
    const size_t lanes = (n + 1 + ZERO_CHUNK - 1) / ZERO_CHUNK;
    hipLaunchKernelGGL((k_prefix_zerofier<P>), dim3(grid256(lanes)), dim3(256), 0, s, (const u32*)d_params.p, (int)m, n + 1, d_z.w());
    MZK_HIP(hipGetLastError());
    z.resize((n + 1) * nl);
    MZK_HIP(hipMemcpyAsync(z.data(), d_z.p, (n + 1) * esz, hipMemcpyDeviceToHost, s));
  }
  MZK_TRY(d2h_sync(&flag, d_flag.p, 4, s));                   // (`params`, `z` and the caller's domain are in flight until here)
  settle.armed = false;
  if (flag) return MZK_OK;                                    // not a prefix after all: the caller runs the tree
  memset(out, 0, len * esz);
  if (m == 0) {                                               // the whole subgroup: X^N - 1
    uint64_t one[4] = {1, 0, 0, 0}, zero[4] = {0, 0, 0, 0}, neg[4] = {0, 0, 0, 0};
    hp_sub(hf, neg, zero, one);
    memcpy(out, neg, 8 * nl);
    out[N * nl] = 1;
  } else memcpy(out, z.data(), (n + 1) * esz);
  *done = true;
  return MZK_OK;
}

// ---- interpolation plans: what fast_interpolate derives from the domain alone ------------------------------------------------------
struct InterpPlanBase {
  int fid = -1;
  size_t n = 0;
  std::vector<uint64_t> domain;      // exact host copy: the key
  uint64_t stamp = 0;
  size_t bytes = 0;
  virtual ~InterpPlanBase() {}
};
template <class P> struct InterpPlan : InterpPlanBase {
  PolyTree<P> T;
  DevBuf d_dom, d_zp;        // the domain; 1 / Z'(d_i)
  // a prefix of a power-of-two subgroup (see k_prefix_weights): no tree; the generator, the subgroup's order, the number of its points
  // that are missing and the weights that give the interpolant's values there
  bool prefix = false;
  uint64_t g[4] = {0, 0, 0, 0};
  size_t pN = 0, pm = 0;
  DevBuf d_C;
};
static std::vector<InterpPlanBase*> g_interp_plans[MZK_MAX_CTX];
static InterpPlanBase* g_interp_busy[MZK_MAX_CTX];     // the cached plan the running call works with: never evicted under it
static uint64_t g_interp_stamp = 0;
constexpr size_t INTERP_MAX_PLANS = 4;
constexpr size_t INTERP_MAX_BYTES = (size_t)1 << 30;
void poly_release_plans() {
  auto& v = g_interp_plans[ctx().index];
  for (auto* p : v) delete p;
  v.clear();
  g_interp_busy[ctx().index] = nullptr;
}
// Device bytes this file holds on the current context outside any call: blocks parked in the pool + the cached plans (their own
// estimate).  Part of mzk_workspace_bytes and of the workspace budget.
size_t poly_bytes_held() {
  size_t t = g_poly_pool[ctx().index].bytes;
  for (auto* b : g_interp_plans[ctx().index]) t += b->bytes;
  return t;
}
// Everything here the RUNNING call does not use goes back to the device: the cached plans but the busy one (their buffers pass
// through the pool), then the parked blocks.  Returns the bytes released.  Called when an allocation was refused or would exceed
// the budget (ws_trim_idle) and by mzk_trim_workspace.
size_t poly_trim_idle() {
  const size_t before = poly_bytes_held();
  auto& v = g_interp_plans[ctx().index];
  InterpPlanBase* busy = g_interp_busy[ctx().index];
  bool any = false;
  for (size_t i = v.size(); i-- > 0;) {
    if (v[i] == busy) continue;
    if (!any) { (void)hipDeviceSynchronize(); any = true; }      // a plan's buffers may still be read by enqueued work of an earlier call
    delete v[i];
    v.erase(v.begin() + (long)i);
  }
  poly_release_free_blocks();
  const size_t after = poly_bytes_held();
  return before > after ? before - after : 0;
}
// the domain of an interpolation is canonical if a cached plan was built from exactly these limbs (it was checked then); otherwise check it now
static int interp_check_domain(int fid, const HostField* hf, const uint64_t* domain, size_t n) {
  for (auto* b : g_interp_plans[ctx().index])
    if (b->fid == fid && b->n == n && n && !memcmp(b->domain.data(), domain, n * (size_t)hf->nl * 8)) return MZK_OK;
  return check_canonical(hf, domain, n, "domain");
}
template <class P>
static int interp_plan_get(int fid, const uint64_t* domain, size_t n, const uint64_t* root, size_t root_order, hipStream_t s, InterpPlan<P>** out, bool* transient) {
  *transient = false;
  auto& v = g_interp_plans[ctx().index];
  const size_t nl = (size_t)host_field(fid)->nl;
  for (auto* b : v) {
    if (b->fid == fid && b->n == n && !memcmp(b->domain.data(), domain, n * nl * 8)) {
      b->stamp = ++g_interp_stamp;
      *out = static_cast<InterpPlan<P>*>(b);
      return MZK_OK;
    }
  }
  const size_t esz = field_bytes(fid);
  const HostField* hf = host_field(fid);
  InterpPlan<P>* pl = new InterpPlan<P>();
  pl->fid = fid; pl->n = n;
  int rc = tree_init(&pl->T, fid, n, root, root_order, s);
  DevBuf d_dz, d_zpv;
  if (rc == MZK_OK) rc = pl->d_dom.alloc(n * esz);
  if (rc == MZK_OK && hipMemcpyAsync(pl->d_dom.p, domain, n * esz, hipMemcpyHostToDevice, s) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipMemcpyAsync", __FILE__, __LINE__);
  std::vector<uint64_t> ainv;
  if (rc == MZK_OK && prefix_candidate(hf, domain, n, &pl->pN)) {
    DevBuf d_flag;
    u32 flag = 1;
    rc = d_flag.alloc(4);
    if (rc == MZK_OK) rc = prefix_check_launch<P>(pl->d_dom.p, n, (u32*)d_flag.p, s);
    if (rc == MZK_OK) rc = d2h_sync(&flag, d_flag.p, 4, s);
    if (rc == MZK_OK && flag == 0) {
      pl->pm = pl->pN - n;
      memcpy(pl->g, domain + nl, 8 * nl);
      pl->prefix = pl->pm == 0 || prefix_system_inverse(hf, pl->g, n, pl->pm, &ainv);
    }
  }
  if (rc == MZK_OK && pl->prefix && pl->pm) {
    const size_t m = pl->pm;
    DevBuf d_ainv;
    rc = d_ainv.alloc(m * m * esz);
    if (rc == MZK_OK) rc = pl->d_C.alloc(m * n * esz);
    if (rc == MZK_OK && hipMemcpyAsync(d_ainv.p, ainv.data(), m * m * esz, hipMemcpyHostToDevice, s) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipMemcpyAsync", __FILE__, __LINE__);
    if (rc == MZK_OK) {
      hipLaunchKernelGGL((k_prefix_weights<P>), dim3(grid256(n), (unsigned)m), dim3(256), 0, s, (const u32*)pl->d_dom.p, n, (const u32*)d_ainv.p, (int)m, pl->d_C.w());
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = hip_fail(hipGetLastError(), "k_prefix_weights", __FILE__, __LINE__);      // `ainv`, d_ainv are read until here
    }
  }
  if (rc == MZK_OK && !pl->prefix) rc = d_dz.alloc(n * esz);
  if (rc == MZK_OK && !pl->prefix) rc = pl->d_zp.alloc(pl->T.N * esz);
  if (rc == MZK_OK && !pl->prefix) rc = pl->T.build(pl->d_dom.p, true);
  if (rc == MZK_OK && !pl->prefix) {
    // w_i = v_i / Z'(d_i); a repeated point has Z' = 0 and the reference's division by inverse(0) = 0 (field.rs:209-232)
    // zeroes its target at the level that separates the two copies (ntt.rs:233-242): w_i = 0 as well
    hipLaunchKernelGGL((k_derivative<P>), dim3(grid256(n)), dim3(256), 0, s, (const u32*)pl->T.low[pl->T.levels].w(), pl->T.N, pl->T.pad, n, d_dz.w());
    rc = d_zpv.alloc(pl->T.N * esz);
    if (rc == MZK_OK) rc = pl->T.evaluate(d_dz.p, n, pl->d_dom.p, d_zpv.p, n);
  }
  if (rc == MZK_OK && !pl->prefix) {
    // the plan keeps 1 / Z'(d_i) (0 where Z' = 0, as the division gives): d_dz is free again and takes the numerators, all ones
    std::vector<uint64_t> ones(n * nl, 0);
    for (size_t i = 0; i < n; i++) ones[i * nl] = 1;
    if (hipMemcpyAsync(d_dz.p, ones.data(), n * esz, hipMemcpyHostToDevice, s) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipMemcpyAsync", __FILE__, __LINE__);
    if (rc == MZK_OK) rc = pointwise_div_shared_dev(fid, d_dz.p, n, d_zpv.p, pl->d_zp.p, n, n, 1, s);
    if (hipStreamSynchronize(s) != hipSuccess && rc == MZK_OK) rc = hip_fail(hipGetLastError(), "hipStreamSynchronize", __FILE__, __LINE__);      // `ones` is read until here
  }
  if (rc != MZK_OK) { (void)hipStreamSynchronize(s); delete pl; return rc; }
  pl->domain.assign(domain, domain + n * nl);
  pl->stamp = ++g_interp_stamp;
  pl->bytes = pl->prefix ? (pl->pm + 1) * n * esz : (size_t)(3 * pl->T.levels + 4) * pl->T.N * esz;
  size_t total = pl->bytes;
  for (auto* b : v) total += b->bytes;
  while (pl->bytes <= INTERP_MAX_BYTES && !v.empty() && (v.size() >= INTERP_MAX_PLANS || total > INTERP_MAX_BYTES)) {      // least recently used first; their buffers return to the pool
    size_t victim = 0;
    for (size_t i = 1; i < v.size(); i++) if (v[i]->stamp < v[victim]->stamp) victim = i;
    (void)hipStreamSynchronize(s);
    total -= v[victim]->bytes;
    delete v[victim];
    v.erase(v.begin() + (long)victim);
  }
  if (pl->bytes <= INTERP_MAX_BYTES) v.push_back(pl);
  else *transient = true;           // too large to keep (2^22 points and up): the caller deletes it after use, as before round 5
  *out = pl;
  return MZK_OK;
}

template <class P>
// values / out on the host, or -- d_values / d_out non-null -- in HBM (mzk_fast_interpolate_batch_dev: `user` is the stream the caller's
// buffers are ordered on; the work runs on the context's stream behind an event and is complete when the call returns)
static int interpolate_impl(int fid, const uint64_t* domain, const uint64_t* values, size_t n, size_t batch, const uint64_t* root, size_t root_order,
                            uint64_t* out, size_t* out_lens, const void* d_values = nullptr, void* d_out = nullptr, hipStream_t user = nullptr) {
  const HostField* hf = host_field(fid);
  const int nl = hf->nl;
  const size_t esz = field_bytes(fid);
  if (n == 0) { for (size_t r = 0; r < batch; r++) out_lens[r] = 0; return MZK_OK; }                 // ntt.rs:207-209
  const bool on_device = d_values != nullptr;
  if (n == 1) {                                                                                      // ntt.rs:211-215: coef = [values[0]], untrimmed
    if (on_device) {
      MZK_HIP(hipMemcpyAsync(d_out, d_values, batch * esz, hipMemcpyDeviceToDevice, user));
      MZK_HIP(hipStreamSynchronize(user));
    } else memcpy(out, values, batch * esz);
    for (size_t r = 0; r < batch; r++) out_lens[r] = 1;
    return MZK_OK;
  }
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  if (on_device && user != s) {          // the caller's buffers are complete on `user`: the context's stream waits for that point
    hipEvent_t ev;
    MZK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e1 = hipEventRecord(ev, user), e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, ev, 0) : e1;
    (void)hipEventDestroy(ev);
    MZK_HIP(e2);
  }
  MZK_TRY(check_order_for(n - n / 2, root_order, "fast_interpolate"));        // ntt.rs:219-220
  // Everything that depends on the DOMAIN alone -- the subproduct tree with its transformed levels, Z'(d_i) -- is a plan, kept per
  // context like the twiddle tables of a transform: a STARK prover interpolates every trace over the same omicron domain
  // (fast_stark.rs:209-229), and that half of the work (zerofier tree, Newton inverse, the whole down-sweep) is the same each time.
  // Keyed by the exact domain (a host copy is compared), not by a hash.
  InterpPlan<P>* plan = nullptr;
  bool transient = false;
  MZK_TRY(interp_plan_get<P>(fid, domain, n, root, root_order, s, &plan, &transient));
  struct Drop {         // a plan too large for the cache lives for this call only (every exit path; the stream is idle by then or is waited for)
    InterpPlanBase* p; hipStream_t s; int idx;
    ~Drop() { g_interp_busy[idx] = nullptr; if (p) { (void)hipStreamSynchronize(s); delete p; } }
  } drop{transient ? plan : nullptr, s, ctx().index};
  g_interp_busy[ctx().index] = transient ? nullptr : plan;      // an allocation of this call that runs out of memory evicts the OTHER cached plans only
  if (plan->prefix) {
    // values -> [values | the interpolant at the missing points] -> inverse transform with the domain's generator: coefficients
    const size_t N = plan->pN, m = plan->pm;
    int lgN = 0;
    while (((size_t)1 << lgN) < N) lgN++;
    const size_t group = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(batch, ((size_t)1 << 24) / N), 65535));      // (65535: the register index is a grid's y)
    DevBuf d_vals, d_ext, d_lens;
    if (!on_device) MZK_TRY(d_vals.alloc(group * n * esz));
    MZK_TRY(d_ext.alloc(group * N * esz)); MZK_TRY(d_lens.alloc(group * 4 + 64));
    std::vector<u32> lens(group);
    for (size_t r0 = 0; r0 < batch; r0 += group) {
      const size_t g = std::min(group, batch - r0);
      const void* vals = on_device ? (const void*)((const uint8_t*)d_values + r0 * n * esz) : d_vals.p;
      if (!on_device) MZK_HIP(hipMemcpyAsync(d_vals.p, values + r0 * n * nl, g * n * esz, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL((k_prefix_pad<P>), dim3(grid256(g * N)), dim3(256), 0, s, (const u32*)vals, n, lgN, g * N, d_ext.w());
      if (m) hipLaunchKernelGGL((k_prefix_missing<P>), dim3((unsigned)m, (unsigned)g), dim3(PREFIX_NT), 0, s, (const u32*)vals, n, N, (const u32*)plan->d_C.p, d_ext.w());
      MZK_HIP(hipGetLastError());
      MZK_TRY(ntt_batch_dev_impl(fid, plan->g, d_ext.p, d_ext.p, N, g, 1, s));
      hipLaunchKernelGGL((k_row_len<P>), dim3((unsigned)g), dim3(256), 0, s, (const u32*)d_ext.p, N, (size_t)0, n, (u32*)d_lens.p);
      MZK_HIP(hipGetLastError());
      if (on_device) MZK_HIP(hipMemcpy2DAsync((uint8_t*)d_out + r0 * n * esz, n * esz, d_ext.p, N * esz, n * esz, g, hipMemcpyDeviceToDevice, s));
      else MZK_HIP(hipMemcpy2DAsync(out + r0 * n * nl, n * esz, d_ext.p, N * esz, n * esz, g, hipMemcpyDeviceToHost, s));
      MZK_TRY(d2h_sync(lens.data(), d_lens.p, g * 4, s));
      for (size_t r = 0; r < g; r++) out_lens[r0 + r] = lens[r];
    }
    return MZK_OK;
  }
  PolyTree<P>& T = plan->T;
  DevBuf& d_dom = plan->d_dom;
  DevBuf& d_zp = plan->d_zp;
  // registers go through the up-sweep in groups (at most 2^24 elements per buffer)
  const size_t group = std::max<size_t>(1, std::min<size_t>(batch, ((size_t)1 << 24) / T.N));
  DevBuf d_vals, d_ws, d_ress, d_lens;
  if (!on_device) MZK_TRY(d_vals.alloc(group * n * esz));
  MZK_TRY(d_ws.alloc(group * T.N * esz)); MZK_TRY(d_ress.alloc(group * T.N * esz)); MZK_TRY(d_lens.alloc(group * 4 + 64));
  std::vector<u32> lens(group);
  for (size_t r0 = 0; r0 < batch; r0 += group) {
    const size_t g = std::min(group, batch - r0);
    const void* vals = on_device ? (const void*)((const uint8_t*)d_values + r0 * n * esz) : d_vals.p;
    if (!on_device) MZK_HIP(hipMemcpyAsync(d_vals.p, values + r0 * n * nl, g * n * esz, hipMemcpyHostToDevice, s));
    MZK_HIP(hipMemsetAsync(d_ws.p, 0, g * T.N * esz, s));
    // all registers of the group in one launch: they divide by the same Z'(d_i), whose inverses the plan holds (round 4: one launch and
    // one inversion chain per REGISTER, 16 x 60 us of a 3.5-ms batch of 16 registers of 2^14 points; then one shared chain, 0.18 ms)
    MZK_TRY(pointwise_mul_shared_dev(fid, vals, n, d_zp.p, d_ws.p, T.N, n, g, s));
    MZK_TRY(T.combine(d_dom.p, d_ws.p, d_ress.p, g));
    // sum_i w_i Z_pad / (X - d_i) = X^pad * interpolant: row r of the result is elements [pad, pad + n) of its T.N-element row, copied out as
    // n elements (zeros beyond the trimmed length: polynomial.rs:214-228 trims like the final `+`; the length comes from the device)
    hipLaunchKernelGGL((k_row_len<P>), dim3((unsigned)g), dim3(256), 0, s, (const u32*)d_ress.p, T.N, T.pad, n, (u32*)d_lens.p);
    MZK_HIP(hipGetLastError());
    const uint8_t* src = (const uint8_t*)d_ress.p + T.pad * esz;
    if (on_device) MZK_HIP(hipMemcpy2DAsync((uint8_t*)d_out + r0 * n * esz, n * esz, src, T.N * esz, n * esz, g, hipMemcpyDeviceToDevice, s));
    else MZK_HIP(hipMemcpy2DAsync(out + r0 * n * nl, n * esz, src, T.N * esz, n * esz, g, hipMemcpyDeviceToHost, s));
    MZK_TRY(d2h_sync(lens.data(), d_lens.p, g * 4, s));
    for (size_t r = 0; r < g; r++) out_lens[r0 + r] = lens[r];
  }
  return MZK_OK;
}

}  // namespace mzk

using namespace mzk;

extern "C" {

int mzk_fast_zerofier(int field_id, const uint64_t* domain, size_t n, const uint64_t* root, size_t root_order, uint64_t* out, size_t* out_len) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("fast_zerofier: bad field id %d", field_id); return MZK_E_ARG; }
  if (!root || !out_len || (n && (!domain || !out))) { set_error("fast_zerofier: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(field_id);
  MZK_TRY(check_root(hf, root, root_order));
  MZK_TRY(check_canonical(hf, domain, n, "domain"));
  return field_id == MZK_FIELD_M128 ? zerofier_impl<M128Params>(field_id, domain, n, root, root_order, out, out_len)
                                    : zerofier_impl<FrParams>(field_id, domain, n, root, root_order, out, out_len);
}
int mzk_fast_evaluate(int field_id, const uint64_t* coef, size_t m, const uint64_t* domain, size_t n, const uint64_t* root, size_t root_order, uint64_t* out) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("fast_evaluate: bad field id %d", field_id); return MZK_E_ARG; }
  if (!root || (m && !coef) || (n && (!domain || !out))) { set_error("fast_evaluate: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(field_id);
  MZK_TRY(check_root(hf, root, root_order));
  MZK_TRY(check_canonical(hf, domain, n, "domain"));
  MZK_TRY(check_canonical(hf, coef, m, "coef"));
  return field_id == MZK_FIELD_M128 ? evaluate_impl<M128Params>(field_id, coef, m, domain, n, root, root_order, out)
                                    : evaluate_impl<FrParams>(field_id, coef, m, domain, n, root, root_order, out);
}
int mzk_fast_interpolate(int field_id, const uint64_t* domain, const uint64_t* values, size_t n, const uint64_t* root, size_t root_order, uint64_t* out,
                         size_t* out_len) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("fast_interpolate: bad field id %d", field_id); return MZK_E_ARG; }
  if (!root || !out_len || (n && (!domain || !values || !out))) { set_error("fast_interpolate: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(field_id);
  MZK_TRY(check_root(hf, root, root_order));
  MZK_TRY(interp_check_domain(field_id, hf, domain, n));
  MZK_TRY(check_canonical(hf, values, n, "values"));
  return field_id == MZK_FIELD_M128 ? interpolate_impl<M128Params>(field_id, domain, values, n, 1, root, root_order, out, out_len)
                                    : interpolate_impl<FrParams>(field_id, domain, values, n, 1, root, root_order, out, out_len);
}
int mzk_fast_interpolate_batch(int field_id, const uint64_t* domain, const uint64_t* values, size_t n, size_t batch, const uint64_t* root,
                               size_t root_order, uint64_t* out, size_t* out_lens) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("fast_interpolate: bad field id %d", field_id); return MZK_E_ARG; }
  if (batch == 0) return MZK_OK;
  if (!root || !out_lens || (n && (!domain || !values || !out))) { set_error("fast_interpolate: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(field_id);
  MZK_TRY(check_root(hf, root, root_order));
  MZK_TRY(interp_check_domain(field_id, hf, domain, n));
  MZK_TRY(check_canonical(hf, values, n * batch, "values"));
  return field_id == MZK_FIELD_M128 ? interpolate_impl<M128Params>(field_id, domain, values, n, batch, root, root_order, out, out_lens)
                                    : interpolate_impl<FrParams>(field_id, domain, values, n, batch, root, root_order, out, out_lens);
}

int mzk_fast_interpolate_batch_dev(int field_id, const uint64_t* domain, const void* d_values, size_t n, size_t batch, const uint64_t* root,
                                   size_t root_order, void* d_out, size_t* out_lens, void* stream) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("fast_interpolate: bad field id %d", field_id); return MZK_E_ARG; }
  if (batch == 0) return MZK_OK;
  if (!root || !out_lens || (n && (!domain || !d_values || !d_out))) { set_error("fast_interpolate: null pointer"); return MZK_E_ARG; }
  const HostField* hf = host_field(field_id);
  MZK_TRY(check_root(hf, root, root_order));
  MZK_TRY(interp_check_domain(field_id, hf, domain, n));
  return field_id == MZK_FIELD_M128 ? interpolate_impl<M128Params>(field_id, domain, nullptr, n, batch, root, root_order, nullptr, out_lens, d_values, d_out, (hipStream_t)stream)
                                    : interpolate_impl<FrParams>(field_id, domain, nullptr, n, batch, root, root_order, nullptr, out_lens, d_values, d_out, (hipStream_t)stream);
}

}  // extern "C"
