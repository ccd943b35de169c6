// One kernel per elementary operation of the M128 / Fr transform kernels (mzk_ntt.hip), so that the instruction cost of each can be
// read off the ISA: tools/ntt_instruction_budget.py compiles this file for gfx950, disassembles it, and subtracts the `base` kernel
// (the same loads and stores with no arithmetic) from each.  Never linked into the library.
#include "../../myzkp_amd/csrc/mzk_common.h"
#include "../../myzkp_amd/csrc/mzk_field_asm.h"
using namespace mzk;

template <class P> __device__ __forceinline__ Fe<P> ld(const u32* p, int i) {
  Fe<P> r;
#pragma unroll
  for (int k = 0; k < P::L; k++) r.l[k] = p[(size_t)(i * P::L + k) * 64];
  return r;
}
template <class P> __device__ __forceinline__ void st(u32* p, int i, const Fe<P>& v) {
#pragma unroll
  for (int k = 0; k < P::L; k++) p[(size_t)(i * P::L + k) * 64] = v.l[k];
}
#define OPK(name, P, ...)                                                               \
  extern "C" __global__ void name(const u32* __restrict__ in, u32* __restrict__ out) {  \
    in += threadIdx.x; out += threadIdx.x;                                              \
    Fe<P> a = ld<P>(in, 0), b = ld<P>(in, 1);                                           \
    Fe<P> r0 = a, r1 = b;                                                               \
    __VA_ARGS__;                                                                            \
    st<P>(out, 0, r0); st<P>(out, 1, r1);                                               \
  }
OPK(m128_base, M128Params, {})
OPK(m128_smul, M128Params, { r0 = FeAsm<M128Params>::smul(a, b); })
OPK(m128_mul, M128Params, { r0 = FeAsm<M128Params>::mul(a, b); })
OPK(m128_sbfly, M128Params, { r0 = fe_sadd<M128Params>(a, b); r1 = fe_ssub<M128Params>(a, b); })
OPK(m128_scarry, M128Params, { r0 = fe_scarry<M128Params>(a); })
OPK(m128_sbias, M128Params, { r0 = fe_sbias<M128Params>(a); })
OPK(m128_sreduce, M128Params, { r0 = fe_sreduce<M128Params>(a); })
OPK(m128_bfly_lazy_r4, M128Params, { r1 = fe_sub<M128Params, 8>(a, b); r0 = fe_add<M128Params>(a, b); })
OPK(m128_bfly_carry_r4, M128Params, { r1 = fe_sub_carry<M128Params, 8>(a, b); r0 = fe_add_carry<M128Params>(a, b); })
OPK(m128_reduce_r4, M128Params, { r0 = fe_reduce<M128Params>(a); })
OPK(m128_weak_reduce_r4, M128Params, { r0 = fe_weak_reduce<M128Params>(a); })
OPK(fr_base, FrParams, {})
OPK(fr_mul, FrParams, { r0 = FeAsm<FrParams>::mul(a, b); })
OPK(fr_bfly_lazy, FrParams, { r1 = fe_sub<FrParams, 8>(a, b); r0 = fe_add<FrParams>(a, b); })
OPK(fr_bfly_carry, FrParams, { r1 = fe_sub_carry<FrParams, 8>(a, b); r0 = fe_add_carry<FrParams>(a, b); })
OPK(fr_reduce, FrParams, { r0 = fe_reduce<FrParams>(a); })
OPK(fr_weak_reduce, FrParams, { r0 = fe_weak_reduce<FrParams>(a); })
OPK(fr_fit, FrParams, { r0 = fe_cond_sub_p<FrParams>(a); })
// pack / unpack: words in, limbs out and back
extern "C" __global__ void m128_unpack(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  u32 w[4];
  for (int k = 0; k < 4; k++) w[k] = in[k * 64];
  st<M128Params>(out, 0, fe_unpack<M128Params>(w));
}
extern "C" __global__ void m128_unpack_base(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  Fe<M128Params> r;
  for (int k = 0; k < 4; k++) r.l[k] = in[k * 64];
  r.l[4] = 0;
  st<M128Params>(out, 0, r);
}
extern "C" __global__ void m128_pack(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  u32 w[4];
  fe_pack<M128Params>(ld<M128Params>(in, 0), w);
  for (int k = 0; k < 4; k++) out[k * 64] = w[k];
}
extern "C" __global__ void m128_pack_base(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  const Fe<M128Params> a = ld<M128Params>(in, 0);
  for (int k = 0; k < 4; k++) out[k * 64] = a.l[k] + (k == 3 ? a.l[4] : 0u);
}
extern "C" __global__ void fr_unpack(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  u32 w[8];
  for (int k = 0; k < 8; k++) w[k] = in[k * 64];
  st<FrParams>(out, 0, fe_unpack<FrParams>(w));
}
extern "C" __global__ void fr_unpack_base(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  Fe<FrParams> r;
  for (int k = 0; k < 8; k++) r.l[k] = in[k * 64];
  r.l[8] = 0;
  st<FrParams>(out, 0, r);
}
extern "C" __global__ void fr_pack(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  u32 w[8];
  fe_pack<FrParams>(ld<FrParams>(in, 0), w);
  for (int k = 0; k < 8; k++) out[k * 64] = w[k];
}
extern "C" __global__ void fr_pack_base(const u32* __restrict__ in, u32* __restrict__ out) {
  in += threadIdx.x; out += threadIdx.x;
  const Fe<FrParams> a = ld<FrParams>(in, 0);
  for (int k = 0; k < 8; k++) out[k * 64] = a.l[k] + (k == 7 ? a.l[8] : 0u);
}
// the Shoup product with scalar constants
extern "C" __global__ void fr_shoup(const u32* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ tab) {
  in += threadIdx.x; out += threadIdx.x;
  Fe<FrParams> a = ld<FrParams>(in, 0), b = ld<FrParams>(in, 1);
  u32 w[9], wq[9];
  for (int i = 0; i < 9; i++) { w[i] = tab[i]; wq[i] = tab[16 + i]; }
  st<FrParams>(out, 0, FeAsm<FrParams>::shoup_mul(a, w, wq)); st<FrParams>(out, 1, b);
}
