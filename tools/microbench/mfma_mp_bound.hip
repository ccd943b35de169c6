// Gated experiment: can the CONSTANT half of a Montgomery product, m * p (81 of the 162 multiply-adds), run
// on the matrix pipe as an int8 GEMM -- Toeplitz(digits of p) x [digits of m, one column per lane] with V_MFMA_I32_16X16X64_I8 --
// beside the VALU products of other waves?  Microbenchmark only, like batched_affine_bound.hip: it prices the pieces.
//
// One wave-product (64 lanes = 64 instances of m, 9 x 29-bit limbs each) needs
//   VALU  a) 9 limbs -> 33 byte digits, biased by 0x80 so that the signed int8 view is exact (constant correction)     ~34 ops
//         b) MFMA wants instance n's digits spread over lanes n, n+16, n+32, n+48 (B operand: K runs over lane rows), the field
//            code has instance n in lane n: a 4 x 4 row transpose per group of four VGPRs, v_permlane16/32_swap            ~16 ops
//   MFMA  c) 4 column blocks (16 instances each) x 5 row blocks (65 byte columns -> 80) = 20 x V_MFMA_I32_16X16X64_I8
//   VALU  d) the 80 accumulator VGPRs hold instance n's column sums spread over four lane rows again: 20 more transposes   ~80 ops
//         e) 65 signed 20-bit byte-column sums -> 9 x 29-bit limbs: sign-extend, shift, 64-bit add each, then carry       ~157 ops
// against the 81 v_mad_u64_u32 + 17 shift / mask it replaces (98 ops).  This file measures the issue cost of each instruction
// kind (SIMD cycles per wave-instruction at 1 and 4 waves per SIMD), the MFMA's issue interval, and whether MFMAs really
// co-issue with a dependent v_mad_u64_u32 stream of ANOTHER wave (they do) -- the verdict is then arithmetic, printed at the end.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_mp_bound.hip -o mfma_mp_bound
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef int v4i __attribute__((ext_vector_type(4)));

extern "C" __global__ void k_mad(u64* out, int iters, u32 sa) {          // 32 dependent-per-accumulator v_mad_u64_u32 per round
  u64 acc[8]; u32 a = threadIdx.x + sa, b = threadIdx.x * 7 + 3;
  for (int i = 0; i < 8; i++) acc[i] = i + sa + threadIdx.x;
  for (int k = 0; k < iters; k++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
  u64 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" __global__ void k_swap16(u64* out, int iters, u32 sa) {       // 32 v_permlane16_swap per round (4 independent pairs)
  u32 v[8]; for (int i = 0; i < 8; i++) v[i] = threadIdx.x * (i + 3) + sa;
  for (int k = 0; k < iters; k++)
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i += 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[i + 1]));
  u64 s = 0; for (int i = 0; i < 8; i++) s ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" __global__ void k_swap32(u64* out, int iters, u32 sa) {
  u32 v[8]; for (int i = 0; i < 8; i++) v[i] = threadIdx.x * (i + 3) + sa;
  for (int k = 0; k < iters; k++)
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[i + 1]));
  u64 s = 0; for (int i = 0; i < 8; i++) s ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" __global__ void k_lshladd64(u64* out, int iters, u32 sa) {    // the recombination's 64-bit shift-add
  u64 acc[8]; u64 b = threadIdx.x * 7 + sa;
  for (int i = 0; i < 8; i++) acc[i] = i + sa + threadIdx.x;
  for (int k = 0; k < iters; k++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_lshl_add_u64 %0, %1, 5, %0" : "+v"(acc[i]) : "v"(b));
  u64 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 32 MFMAs per round on 4 independent accumulators; MIX = 1: waves with an odd index run the v_mad stream instead (co-issue test)
template <int MIX> __global__ __launch_bounds__(256) void k_mfma(u64* out, int iters, u32 sa) {
  const bool mad_wave = MIX && ((threadIdx.x >> 6) & 1);
  v4i acc[4]; v4i a, b;
  for (int i = 0; i < 4; i++) { a[i] = threadIdx.x * (i + 1) + sa; b[i] = threadIdx.x * (i + 5) + 1; acc[i] = v4i{0, 0, 0, 0}; }
  u64 macc[8]; u32 ma = threadIdx.x + sa, mb = threadIdx.x * 7 + 3;
  for (int i = 0; i < 8; i++) macc[i] = i + sa;
  if (!mad_wave) {
    for (int k = 0; k < iters; k++)
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  } else {
    for (int k = 0; k < iters; k++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(macc[i]) : "v"(ma), "v"(mb) : "vcc");
  }
  u64 s = 0;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) s ^= (u32)acc[i][j];
  for (int i = 0; i < 8; i++) s ^= macc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  u64* out; CK(hipMalloc(&out, (size_t)ncu * 4 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double cyc[8][2] = {};
  const char* names[6] = {"v_mad_u64_u32", "v_permlane16_swap_b32", "v_permlane32_swap_b32", "v_lshl_add_u64", "v_mfma_i32_16x16x64_i8 (all waves)",
                          "half the waves MFMA, half v_mad_u64_u32"};
  for (int bi = 0; bi < 2; bi++) {
    const int bpc = bi ? 4 : 1, iters = 2000;
    for (int c = 0; c < 6; c++) {
      dim3 g(ncu * bpc), bl(256);
      auto launch = [&](int it) {
        switch (c) {
          case 0: hipLaunchKernelGGL(k_mad, g, bl, 0, 0, out, it, 1u); break;
          case 1: hipLaunchKernelGGL(k_swap16, g, bl, 0, 0, out, it, 1u); break;
          case 2: hipLaunchKernelGGL(k_swap32, g, bl, 0, 0, out, it, 1u); break;
          case 3: hipLaunchKernelGGL(k_lshladd64, g, bl, 0, 0, out, it, 1u); break;
          case 4: hipLaunchKernelGGL(k_mfma<0>, g, bl, 0, 0, out, it, 1u); break;
          default: hipLaunchKernelGGL(k_mfma<1>, g, bl, 0, 0, out, it, 1u); break;
        }
      };
      launch(8); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      // 32 instructions per round and wave; bpc waves per SIMD (256 threads = 4 waves = one per SIMD)
      cyc[c][bi] = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * bpc);
      printf("%-44s waves/SIMD=%d  %8.3f ms  %6.2f SIMD cycles per wave-instruction%s\n", names[c], bpc, ms, cyc[c][bi],
             c == 5 ? " (per instruction of EITHER kind: co-issue if this is about half of the two separate costs)" : "");
    }
  }
  const double mad = cyc[0][1], sw = 0.5 * (cyc[1][1] + cyc[2][1]), la = cyc[3][1], mf = cyc[4][1];
  const double valu_now = 81 * mad + 17 * 2.4;
  const double valu_mfma = 34 * 2.4 + (16 + 80) * sw + 130 * la + 27 * 2.4;
  printf("\nper wave-product at 4 waves/SIMD:  VALU m*p half now %.0f cycles;  MFMA route: %.0f VALU cycles of digit split / transposes / recombination\n"
         "(+ %.0f cycles on the matrix pipe) -- %s\n", valu_now, valu_mfma, 20 * mf,
         valu_mfma < valu_now ? "worth a prototype" : "REJECTED: the layout conversion alone costs more VALU time than the multiply-adds it would replace");
  return 0;
}
