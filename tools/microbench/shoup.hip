// Price of a precomputed-quotient (Shoup / Harvey) twiddle product against the Montgomery asm block: same
// dependent-chain harness as ubench2.hip (SIMD cycles per wave-product at 1, 2, 4 waves per SIMD), plus a value check -- the
// Shoup result must be congruent to the Montgomery product by w R.  Microbenchmark, not product code.
//   python tools/microbench/gen_shoup.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../myzkp_amd/csrc shoup.hip -o shoup
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "mzk_field.h"
#include "mzk_field_asm.h"
#include "shoup_asm.h"
using namespace mzk;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef unsigned __int128 u128;

template <int V> __global__ __launch_bounds__(256) void k_chain(const u32* in, u32* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fe<FrParams> x, w, wq, wm;
  for (int i = 0; i < 9; i++) { x.l[i] = in[t * 36 + i]; w.l[i] = in[t * 36 + 9 + i]; wq.l[i] = in[t * 36 + 18 + i]; wm.l[i] = in[t * 36 + 27 + i]; }
  for (int k = 0; k < iters; k++) {
    if (V == 0) x = FeAsm<FrParams>::mul(x, wm);        // x * (w R) / R = x w
    else x = shoup_mul(x, w, wq);
  }
  const Fe<FrParams> c = fe_reduce<FrParams>(x);
  for (int i = 0; i < 9; i++) out[t * 9 + i] = c.l[i];
}

static void to_limbs(const uint64_t v[5], u32* l) {   // v: little-endian 64-bit words of a value < 2^261
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, k = bit / 64, s = bit % 64;
    u128 x = v[k]; if (k + 1 < 5) x |= (u128)v[k + 1] << 64;
    l[i] = (u32)((uint64_t)(x >> s) & (i < 8 ? 0x1fffffffu : 0xffffffffu));
  }
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int ncu = pr.multiProcessorCount, n = ncu * 4 * 256;
  // operands: x random < p-ish (canonical limbs), w random 252-bit, wq = floor(w 2^261 / p), wm = w R mod p: done with python-free
  // big arithmetic on the host through unsigned __int128 long division (small helper below)
  const uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  auto shl_mod = [&](uint64_t* r /*4 words, < p*/, int bits, uint64_t* quo /*5 words or null*/) {   // r = r * 2^bits mod p, quo = floor(r 2^bits / p)
    uint64_t q[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < bits; b++) {
      // q <<= 1; r <<= 1; if r >= p: r -= p, q |= 1
      for (int i = 4; i > 0; i--) q[i] = (q[i] << 1) | (q[i - 1] >> 63);
      q[0] <<= 1;
      uint64_t top = r[3] >> 63;
      for (int i = 3; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
      r[0] <<= 1;
      bool ge = top != 0;
      if (!ge) { ge = true; for (int i = 3; i >= 0; i--) if (r[i] != P[i]) { ge = r[i] > P[i]; break; } }
      if (ge) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r[i] - P[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; } q[0] |= 1; }
    }
    if (quo) for (int i = 0; i < 5; i++) quo[i] = q[i];
  };
  std::vector<u32> h((size_t)n * 36);
  uint64_t s = 0x12345;
  auto rnd = [&]() { s += 0x9e3779b97f4a7c15ULL; uint64_t z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
  for (int t = 0; t < n; t++) {
    uint64_t x[5] = {rnd(), rnd(), rnd(), rnd() >> 4, 0}, w[5] = {rnd(), rnd(), rnd(), rnd() >> 4, 0};
    if (t < 64) {   // distinct twiddles only for the first wave's worth (the long division is slow); the rest repeat them
      uint64_t r1[4] = {w[0], w[1], w[2], w[3]}, q1[5];
      shl_mod(r1, 261, q1);
      to_limbs(w, &h[(size_t)t * 36 + 9]);
      to_limbs(q1, &h[(size_t)t * 36 + 18]);
      uint64_t wm[5] = {r1[0], r1[1], r1[2], r1[3], 0};
      to_limbs(wm, &h[(size_t)t * 36 + 27]);
    } else {
      for (int i = 9; i < 36; i++) h[(size_t)t * 36 + i] = h[(size_t)(t % 64) * 36 + i];
    }
    to_limbs(x, &h[(size_t)t * 36]);
  }
  u32 *in, *o0, *o1;
  CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&o0, (size_t)n * 36)); CK(hipMalloc(&o1, (size_t)n * 36));
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<u32> a((size_t)n * 9), b((size_t)n * 9);
  for (int it : {1, 2, 37}) {
    hipLaunchKernelGGL(k_chain<0>, dim3(ncu * 4), dim3(256), 0, 0, in, o0, it);
    hipLaunchKernelGGL(k_chain<1>, dim3(ncu * 4), dim3(256), 0, 0, in, o1, it);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(a.data(), o0, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o1, b.size() * 4, hipMemcpyDeviceToHost));
    printf("Shoup chain of %2d products == Montgomery chain (canonical values): %s\n", it, a == b ? "yes" : "NO");
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int bpc : {1, 2, 4}) for (int v = 0; v < 2; v++) {
    const int iters = 2000; dim3 g(ncu * bpc), bl(256);
    auto launch = [&](int it) { if (v == 0) hipLaunchKernelGGL(k_chain<0>, g, bl, 0, 0, in, o0, it); else hipLaunchKernelGGL(k_chain<1>, g, bl, 0, 0, in, o1, it); };
    launch(4); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s waves/SIMD=%d %8.3f ms  %8.2f Gmul/s  %7.1f SIMD cycles per wave-product\n", v == 0 ? "Montgomery asm block (214 instr, 162 MADs)" : "Shoup asm block (179 instr, 143 MADs)", bpc,
           ms, (double)iters * g.x * bl.x / (ms * 1e-3) / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * bpc));
  }
  return 0;
}
