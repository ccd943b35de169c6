// One wave alone on its SIMD: does instruction-level parallelism raise its issue rate?  Inline-asm blocks of 64 instructions over CH independent
// registers (CH = 1: every instruction depends on the one before; 2, 4, 8: round robin), for v_xor_b32, v_alignbit_b32 and a DPP move.
//   hipcc --offload-arch=gfx950 -O3 -o lone_wave_ilp lone_wave_ilp.hip && ./lone_wave_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int OP, int CH>
__global__ void k(uint32_t* out, int iters) {
  uint32_t r0 = threadIdx.x + 1, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19, kk = 0x9e3779b9u;
  for (int i = 0; i < iters; i++) {
#define XOR(a) "v_xor_b32 %" #a ", %8, %" #a "\n\t"
#define ALB(a) "v_alignbit_b32 %" #a ", %" #a ", %" #a ", 7\n\t"
#define DPP(a) "v_mov_b32_dpp %" #a ", %" #a " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
#define BLOCK(I) \
    if (CH == 1) asm volatile(REP8(I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(kk)); \
    if (CH == 2) asm volatile(REP8(I(0) I(1) I(0) I(1) I(0) I(1) I(0) I(1)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(kk)); \
    if (CH == 4) asm volatile(REP8(I(0) I(1) I(2) I(3) I(0) I(1) I(2) I(3)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(kk)); \
    if (CH == 8) asm volatile(REP8(I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(kk));
    if (OP == 0) { BLOCK(XOR) }
    if (OP == 1) { BLOCK(ALB) }
    if (OP == 2) { BLOCK(DPP) }
  }
  out[threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
template <int OP, int CH> static void run(const char* name, uint32_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000; float best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0); hipLaunchKernelGGL((k<OP, CH>), dim3(1), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-34s %d independent chain(s): %.3f ns per instruction\n", name, CH, best * 1e6 / ((double)iters * 64));
}
int main() {
  uint32_t* d; hipMalloc(&d, 1024);
  run<0, 1>("v_xor_b32", d); run<0, 2>("v_xor_b32", d); run<0, 4>("v_xor_b32", d); run<0, 8>("v_xor_b32", d);
  run<1, 1>("v_alignbit_b32", d); run<1, 2>("v_alignbit_b32", d); run<1, 4>("v_alignbit_b32", d); run<1, 8>("v_alignbit_b32", d);
  run<2, 1>("v_mov_b32 DPP (+ s_nop 1)", d); run<2, 4>("v_mov_b32 DPP (+ s_nop 1)", d); run<2, 8>("v_mov_b32 DPP (+ s_nop 1)", d);
  return 0;
}
