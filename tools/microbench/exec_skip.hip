// Does a wave whose EXEC mask covers only the first 16 / 32 lanes issue its VALU instructions faster than a full wave?
// One wave, a dependent chain of 32-bit ops (v_alignbit / v_xor / v_bitop3 mix as in Keccak), timed with the wall clock of the stream.
//   hipcc --offload-arch=gfx950 -O3 -o exec_skip exec_skip.hip && ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_chain(uint32_t* out, int active, int iters) {
  const int lane = threadIdx.x;
  uint32_t a = lane * 2654435761u + 1, b = a ^ 0x9e3779b9u, c = a + 7;
  if (lane < active) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        a = __builtin_amdgcn_alignbit(a, b, 7) ^ c;
        b = (b ^ a) + c;
        c = __builtin_amdgcn_alignbit(c, a, 13) ^ b;
      }
    }
    out[lane] = a ^ b ^ c;
  }
}
__global__ void k_chain_mad(uint32_t* out, int active, int iters) {
  const int lane = threadIdx.x;
  uint64_t acc = lane;
  uint32_t a = lane * 2654435761u + 1, b = a ^ 0x9e3779b9u;
  if (lane < active) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        acc = (uint64_t)a * b + acc;
        a = (uint32_t)acc ^ b;
      }
    }
    out[lane] = (uint32_t)acc ^ a;
  }
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int kind = 0; kind < 2; kind++)
    for (int active : {64, 48, 32, 16, 8, 1}) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, active, iters);
        else hipLaunchKernelGGL(k_chain_mad, dim3(1), dim3(64), 0, 0, d, active, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double instrs = (double)iters * 16 * (kind == 0 ? 6 : 2);
      printf("%s chain, %2d active lanes: %.3f ms  = %.2f ns per instruction (~%.1f cycles at 2.4 GHz)\n", kind == 0 ? "alignbit/xor/add" : "mad_u64_u32/xor ", active, best,
             best * 1e6 / instrs, best * 1e6 / instrs * 2.4);
    }
  return 0;
}
