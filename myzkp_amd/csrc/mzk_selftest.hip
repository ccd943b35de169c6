// mzk_selftest.hip -- on-device self-check of the inline-asm field products (mzk_field_asm.h) against the portable C++
// forms of mzk_field.h.  The asm blocks exist only in device code, so the host-compiled bounds-checked unit tests
// (tests/hostcheck) cannot reach them; this kernel feeds both forms the same operands -- random limbs, all-ones limbs,
// zeros, and the widest lazy operands the callers produce (limbs up to 2^30.6 against normalised ones) -- and counts
// the lanes whose results differ in any limb.  Called by tests/test_gpu_field_asm.py through mzk_selftest_field_asm.
#include <type_traits>
#include "mzk_common.h"
#include "mzk_field_asm.h"

namespace mzk {

__device__ __forceinline__ u64 st_mix(u64& s) {
  u64 z = (s += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
// kind 0: normalised random limbs (< 2^29, top limb < 2^29 too); 1: every limb 2^29 - 1; 2: zero;
// 3: lazy limbs up to 2^30.6 (the borrow-friendly K p - b forms); 4: a single non-zero limb
template <class P> __device__ Fe<P> st_operand(u64& s, int kind) {
  Fe<P> r;
  const u32 lazy_max = 0x60000000u;      // 2^30.58
  for (int i = 0; i < P::L; i++) {
    const u32 v = (u32)st_mix(s);
    switch (kind) {
      case 0: r.l[i] = v & MASK29; break;
      case 1: r.l[i] = MASK29; break;
      case 2: r.l[i] = 0; break;
      case 3: r.l[i] = v % lazy_max; break;
      default: r.l[i] = 0; break;
    }
  }
  if (kind == 4) r.l[(u32)st_mix(s) % P::L] = (u32)st_mix(s) & MASK29;
  return r;
}
template <class P> __device__ bool st_same(const Fe<P>& a, const Fe<P>& b) {
  u32 d = 0;
  for (int i = 0; i < P::L; i++) d |= a.l[i] ^ b.l[i];
  return d == 0;
}
template <class P>
__global__ void k_selftest_field_asm(u64 seed, size_t n, unsigned long long* __restrict__ mismatches) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s = seed ^ (i * 0xd1342543de82ef95ULL);
  const int ka = (int)(i % 5), kb = (int)((i / 5) % 3);      // b stays normalised (kinds 0..2): the callers' contract
  const Fe<P> a = st_operand<P>(s, ka), b = st_operand<P>(s, kb);
  const Fe<P> c = st_operand<P>(s, (int)((i / 15) % 5)), d = st_operand<P>(s, (int)((i / 75) % 3));
  int bad = 0;
  bad += !st_same<P>(fe_mul<P>(a, b), FeAsm<P>::mul(a, b));
  const Fe<P> q = (ka == 3) ? b : a;            // squares only ever see carried operands (limbs < 2^30)
  bad += !st_same<P>(fe_sqr<P>(q), FeAsm<P>::sqr(q));
  // the fused pair needs L (|a||b| + |c||d|) + L 2^58 < 2^64: both lazy operands at 2^30.6 fit against normalised ones
  bad += !st_same<P>(fe_mul_add2<P>(a, b, c, d), FeAsm<P>::mul_add2(a, b, c, d));
  if constexpr (std::is_same<P, FrParams>::value) {
    // the precomputed-quotient product of the NTT's wave-uniform twiddles: constants in scalar registers (one pair per wave),
    // x any of the operand kinds incl. the lazy one; both forms compute the same columns whatever the constants are
    u64 sw = seed ^ (0x2545f4914f6cdd1dULL * ((i >> 6) + 1));
    u32 w[P::L], wq[P::L];
    for (int k = 0; k < P::L; k++) {
      w[k] = (u32)__builtin_amdgcn_readfirstlane((int)((u32)st_mix(sw) & MASK29));
      wq[k] = (u32)__builtin_amdgcn_readfirstlane((int)((u32)st_mix(sw) & MASK29));
    }
    Fe<P> x = a;
    x.l[P::L - 1] &= 0x07ffffffu;            // value below 2^261: the quotient must fit its nine limbs
    bad += !st_same<P>(fe_shoup_mul<P>(x, w, wq), FeAsm<P>::shoup_mul(x, w, wq));
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// The memory-bound yardstick: one chunk of 4 x 256 x 16 bytes per workgroup (no grid-stride loop: short-lived workgroups keep more
// requests in flight across the chip than 8 persistent ones per CU), four independent 16-byte loads per lane, non-temporal loads and
// stores (the data is touched once).  tools/microbench/copy_bw.hip measured the shapes on MI355X (profiles/round5_copy_kernel_variants.txt):
// 6.26 TB/s read + write for this one against 4.3-4.7 for grid-stride forms, 5.5 for hipMemcpyAsync, 4.7-5.1 for torch's copy_.
__device__ __forceinline__ uint4 nt_load16(const uint4* p) {
  uint4 v;
  v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
  return v;
}
__device__ __forceinline__ void nt_store16(uint4* p, const uint4& v) {
  __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  uint4 v[4];
#pragma unroll
  for (int u = 0; u < 4; u++) if (base + (size_t)u * 256 < n16) v[u] = nt_load16(src + base + (size_t)u * 256);
#pragma unroll
  for (int u = 0; u < 4; u++) if (base + (size_t)u * 256 < n16) nt_store16(dst + base + (size_t)u * 256, v[u]);
}
int selftest_copy_impl(const void* d_src, void* d_dst, size_t bytes, hipStream_t s) {
  if (!bytes) return MZK_OK;
  const size_t n16 = bytes / 16;
  const size_t blocks = (n16 + 1023) / 1024;
  if (blocks > 0x7fffffffu) { set_error("selftest_copy: at most 2^45 bytes per call"); return MZK_E_ARG; }
  hipLaunchKernelGGL(k_copy16, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, n16);
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}

int selftest_field_asm_impl(int fid, uint64_t seed, size_t n, uint64_t* mismatches_host, hipStream_t s) {
  unsigned long long* d;
  MZK_TRY(ws_get(WS_MISC_A, 8, (void**)&d));
  MZK_HIP(hipMemsetAsync(d, 0, 8, s));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (n) {
    if (fid == MZK_FIELD_FR) hipLaunchKernelGGL(k_selftest_field_asm<FrParams>, dim3(blocks), dim3(256), 0, s, seed, n, d);
    else if (fid == MZK_FIELD_FQ) hipLaunchKernelGGL(k_selftest_field_asm<FqParams>, dim3(blocks), dim3(256), 0, s, seed, n, d);
    else if (fid == MZK_FIELD_M128) hipLaunchKernelGGL(k_selftest_field_asm<M128Params>, dim3(blocks), dim3(256), 0, s, seed, n, d);
    else { set_error("selftest: unknown field id %d", fid); return MZK_E_ARG; }
    MZK_HIP(hipGetLastError());
  }
  unsigned long long h = 0;
  MZK_HIP(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, s));
  MZK_HIP(hipStreamSynchronize(s));
  *mismatches_host = (uint64_t)h;
  return MZK_OK;
}

}  // namespace mzk
