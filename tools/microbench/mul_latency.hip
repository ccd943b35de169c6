// Latency of ONE dependent chain of field products on a lone wave (the regime of the MSM tails): fe_mul (one column
// accumulator: every multiply-add waits for the previous one) against an arrangement with independent column sums.
// Result on MI355X (profiles/r02x_mul_latency.txt): 0.384 vs 0.388 us per product on a lone wave -- a wave64 issues one
// VALU instruction per ~4.3 cycles whether or not it depends on the previous one, so instruction-level parallelism inside
// a wave buys nothing; only fewer instructions (or more waves) shorten a chain.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -o mul_latency mul_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../myzkp_amd/csrc/mzk_field.h"
using namespace mzk;
namespace mzk {
// the latency-oriented arrangement tried here (NOT in the library: it measured the same): all 2L-1 column sums first, each
// in its own accumulator (independent chains), then the Montgomery reduction walking the columns
template <class P> __device__ Fe<P> fe_reduce_columns(u64 (&c)[2 * P::L]) {
  constexpr int L = P::L;
  Fe<P> r;
#pragma unroll
  for (int k = 0; k < L; k++) {
    const u32 m = ((u32)c[k] * P::N0) & MASK29;
#pragma unroll
    for (int j = 0; j < L; j++) if (P::P[j] != 0) c[k + j] = mzk_mad(m, P::P[j], c[k + j]);
    c[k + 1] += c[k] >> W29;
  }
  u64 col = c[L];
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
    r.l[k - L] = (u32)col & MASK29;
    col = (col >> W29) + c[k + 1];
  }
  r.l[L - 1] = (u32)col;
  return r;
}
template <class P> __device__ Fe<P> fe_mul_ilp(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L;
  u64 c[2 * L];
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    u64 col = 0;
#pragma unroll
    for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); i++) col = mzk_mad(a.l[i], b.l[k - i], col);
    c[k] = col;
  }
  c[2 * L - 1] = 0;
  return fe_reduce_columns<P>(c);
}
template <class P> __device__ Fe<P> fe_sqr_ilp(const Fe<P>& a) {
  constexpr int L = P::L;
  u32 a2[L];
  u64 c[2 * L];
#pragma unroll
  for (int i = 0; i < L; i++) a2[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    u64 col = 0;
#pragma unroll
    for (int i = (k < L ? 0 : k - L + 1); 2 * i < k; i++) col = mzk_mad(a2[i], a.l[k - i], col);
    if ((k & 1) == 0) col = mzk_mad(a.l[k / 2], a.l[k / 2], col);
    c[k] = col;
  }
  c[2 * L - 1] = 0;
  return fe_reduce_columns<P>(c);
}
}  // namespace mzk
template <int MODE> __global__ void k_chain(const u32* in, u32* out, int iters) {
  Fe<FqParams> x = fe_unpack<FqParams>(in + threadIdx.x * 8), y = fe_unpack<FqParams>(in + 8 * 64 + threadIdx.x * 8);
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) x = fe_mul<FqParams>(x, y);
    else if (MODE == 1) x = fe_mul_ilp<FqParams>(x, y);
    else if (MODE == 2) x = fe_sqr<FqParams>(x);
    else x = fe_sqr_ilp<FqParams>(x);
  }
  fe_pack<FqParams>(fe_reduce<FqParams>(x), out + threadIdx.x * 8);
}
template <int MODE> static float run(const u32* d_in, u32* d_out, int iters, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  u32 h[16 * 64];
  for (int i = 0; i < 16 * 64; i++) h[i] = (i % 8 == 7) ? 0x1234567u : 0x9e3779b9u * (i + 1);
  u32 *d_in, *d_out;
  hipMalloc(&d_in, sizeof h); hipMalloc(&d_out, 8 * 64 * 4 * 4096);
  hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 2000;
  u32 r0[8], r1[8];
  for (int blocks : {1, 1024, 4096}) {
    float a = run<0>(d_in, d_out, iters, blocks); hipMemcpy(r0, d_out, 32, hipMemcpyDeviceToHost);
    float b = run<1>(d_in, d_out, iters, blocks); hipMemcpy(r1, d_out, 32, hipMemcpyDeviceToHost);
    float c = run<2>(d_in, d_out, iters, blocks), d = run<3>(d_in, d_out, iters, blocks);
    bool same = true; for (int i = 0; i < 8; i++) same = same && r0[i] == r1[i];
    printf("%4d wave(s): fe_mul %.3f us/product, fe_mul_ilp %.3f us/product (same result: %d); fe_sqr %.3f, fe_sqr_ilp %.3f\n", blocks,
           a * 1e3 / iters, b * 1e3 / iters, (int)same, c * 1e3 / iters, d * 1e3 / iters);
  }
  return 0;
}
