// mzk_merkle.hip -- SHA3-256 Merkle trees over FRI / STARK codewords, device-resident.
//
// Replaces Merkle::commit / Merkle::open (merkle.rs:15-46) as the provers call them: leaves are
// bincode(FiniteFieldElement) of each codeword element (fri.rs:160-166, fast_stark.rs:64-68) and are NOT hashed
// themselves -- a one-leaf tree commits to the leaf bytes (merkle.rs:17-19), so a level-1 node is
// SHA3(leaf_2i || leaf_2i+1) and every node above is SHA3(left32 || right32).  The leaf bytes are produced on
// the fly from the canonical element (layout: see oracle/mzk_oracle_merkle.c header; unpinned third-party format).
//
// Work shape: n/2 + n/4 + ... one-block Keccak-f[1600] permutations (each message fits the 136-byte rate), pure
// 64-bit logic ops -- ALU-bound, ~5k VALU ops per hash; algorithmic HBM traffic is S*n bytes in, 32*(n-1) out.
#include <algorithm>
#include <atomic>
#include "mzk_common.h"
#include "mzk_keccak_asm.h"

namespace mzk {

typedef uint64_t u64;
typedef uint8_t u8;

template <int R> __device__ __forceinline__ u64 rotl64(u64 x) {
  if constexpr (R == 0) return x;
  else return (x << R) | (x >> (64 - R));
}
__constant__ u64 KECCAK_RC[32] = {      // 24 round constants (+ 8 pad words: keccak_f_pair fetches one round ahead)
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// Keccak-f[1600], FIPS 202 section 3.2/3.3: theta, rho+pi (fused with the theta xor), chi, iota; lane (x, y) = a[x + 5y].
// The 64-bit lanes are handled as explicit 32-bit halves: a rotation is then two v_alignbit_b32 (hipcc turns a 64-bit
// rotate into two 64-bit shifts and an or) and chi's a ^ (~b & c) one v_bitop3_b32 per half (instead of v_bfi + v_xor):
// with the column parities as two three-input xors (v_bitop3_b32 again) 190 instructions per round instead of ~330 -- the tree's upper levels are one hash per lane on a nearly empty GPU,
// i.e. pure instruction latency.
struct KW { u32 lo, hi; };
__device__ __forceinline__ KW kw_xor(KW a, KW b) { return KW{a.lo ^ b.lo, a.hi ^ b.hi}; }
template <int R> __device__ __forceinline__ KW kw_rot(KW x) {
  if constexpr (R == 0) return x;
  else if constexpr (R == 32) return KW{x.hi, x.lo};
  else if constexpr (R < 32) return KW{__builtin_amdgcn_alignbit(x.lo, x.hi, 32 - R), __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - R)};
  else return KW{__builtin_amdgcn_alignbit(x.hi, x.lo, 64 - R), __builtin_amdgcn_alignbit(x.lo, x.hi, 64 - R)};
}
__device__ __forceinline__ KW kw_chi(KW a, KW b, KW c) { return KW{a.lo ^ (~b.lo & c.lo), a.hi ^ (~b.hi & c.hi)}; }
// a ^ b ^ c as one v_bitop3_b32 (truth table 0x96) per half: the five-lane column parities of theta in two steps
__device__ __forceinline__ KW kw_xor3(KW a, KW b, KW c) {
  return KW{(u32)__builtin_amdgcn_bitop3_b32((int)a.lo, (int)b.lo, (int)c.lo, 0x96), (u32)__builtin_amdgcn_bitop3_b32((int)a.hi, (int)b.hi, (int)c.hi, 0x96)};
}
__device__ __forceinline__ void keccak_f(u64 (&st)[25]) {
  KW a[25];
#pragma unroll
  for (int i = 0; i < 25; i++) a[i] = KW{(u32)st[i], (u32)(st[i] >> 32)};
#pragma unroll 1
  for (int rnd = 0; rnd < 24; rnd++) {
    const KW c0 = kw_xor3(kw_xor3(a[0], a[5], a[10]), a[15], a[20]);
    const KW c1 = kw_xor3(kw_xor3(a[1], a[6], a[11]), a[16], a[21]);
    const KW c2 = kw_xor3(kw_xor3(a[2], a[7], a[12]), a[17], a[22]);
    const KW c3 = kw_xor3(kw_xor3(a[3], a[8], a[13]), a[18], a[23]);
    const KW c4 = kw_xor3(kw_xor3(a[4], a[9], a[14]), a[19], a[24]);
    const KW d0 = kw_xor(c4, kw_rot<1>(c1)), d1 = kw_xor(c0, kw_rot<1>(c2)), d2 = kw_xor(c1, kw_rot<1>(c3));
    const KW d3 = kw_xor(c2, kw_rot<1>(c4)), d4 = kw_xor(c3, kw_rot<1>(c0));
    const KW b0 = kw_rot<0>(kw_xor(a[0], d0)), b16 = kw_rot<36>(kw_xor(a[5], d0)), b7 = kw_rot<3>(kw_xor(a[10], d0));
    const KW b23 = kw_rot<41>(kw_xor(a[15], d0)), b14 = kw_rot<18>(kw_xor(a[20], d0));
    const KW b10 = kw_rot<1>(kw_xor(a[1], d1)), b1 = kw_rot<44>(kw_xor(a[6], d1)), b17 = kw_rot<10>(kw_xor(a[11], d1));
    const KW b8 = kw_rot<45>(kw_xor(a[16], d1)), b24 = kw_rot<2>(kw_xor(a[21], d1));
    const KW b20 = kw_rot<62>(kw_xor(a[2], d2)), b11 = kw_rot<6>(kw_xor(a[7], d2)), b2 = kw_rot<43>(kw_xor(a[12], d2));
    const KW b18 = kw_rot<15>(kw_xor(a[17], d2)), b9 = kw_rot<61>(kw_xor(a[22], d2));
    const KW b5 = kw_rot<28>(kw_xor(a[3], d3)), b21 = kw_rot<55>(kw_xor(a[8], d3)), b12 = kw_rot<25>(kw_xor(a[13], d3));
    const KW b3 = kw_rot<21>(kw_xor(a[18], d3)), b19 = kw_rot<56>(kw_xor(a[23], d3));
    const KW b15 = kw_rot<27>(kw_xor(a[4], d4)), b6 = kw_rot<20>(kw_xor(a[9], d4)), b22 = kw_rot<39>(kw_xor(a[14], d4));
    const KW b13 = kw_rot<8>(kw_xor(a[19], d4)), b4 = kw_rot<14>(kw_xor(a[24], d4));
    a[0] = kw_chi(b0, b1, b2); a[1] = kw_chi(b1, b2, b3); a[2] = kw_chi(b2, b3, b4); a[3] = kw_chi(b3, b4, b0); a[4] = kw_chi(b4, b0, b1);
    a[5] = kw_chi(b5, b6, b7); a[6] = kw_chi(b6, b7, b8); a[7] = kw_chi(b7, b8, b9); a[8] = kw_chi(b8, b9, b5); a[9] = kw_chi(b9, b5, b6);
    a[10] = kw_chi(b10, b11, b12); a[11] = kw_chi(b11, b12, b13); a[12] = kw_chi(b12, b13, b14); a[13] = kw_chi(b13, b14, b10); a[14] = kw_chi(b14, b10, b11);
    a[15] = kw_chi(b15, b16, b17); a[16] = kw_chi(b16, b17, b18); a[17] = kw_chi(b17, b18, b19); a[18] = kw_chi(b18, b19, b15); a[19] = kw_chi(b19, b15, b16);
    a[20] = kw_chi(b20, b21, b22); a[21] = kw_chi(b21, b22, b23); a[22] = kw_chi(b22, b23, b24); a[23] = kw_chi(b23, b24, b20); a[24] = kw_chi(b24, b20, b21);
    const u64 rc = KECCAK_RC[rnd];
    a[0].lo ^= (u32)rc;
    a[0].hi ^= (u32)(rc >> 32);
  }
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = ((u64)a[i].hi << 32) | a[i].lo;
}

// ---- the same permutation on a PAIR of lanes --------------------------------------------------------------------------
// The upper levels of a tree are a few hundred hashes at most: one hash per lane leaves the GPU empty and the level's
// time is the instruction latency of one permutation (24 x 190 dependent-issue instructions).  Here the even lane of a
// pair holds the low halves of the 25 lanes of the state and the odd lane the high halves; xors and chi are local, and
// a rotation needs the partner's half of the same word: one quad_perm DPP move + one v_alignbit_b32, the same formula
// on both lanes.  ~125 instructions per round and lane instead of 190 (1.3 x the total work: only used where the level
// is latency-bound).  Both lanes of a pair must be active (a DPP read of a disabled lane returns 0).
__device__ __forceinline__ u32 pair_swap(u32 v) {
  u32 r = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
  asm("" : "+v"(r));       // opaque to the DPP-combine pass (see mzk_coop.h)
  return r;
}
template <int R> __device__ __forceinline__ u32 pair_rot(u32 mine) {
  if constexpr (R == 0) return mine;
  else {
    const u32 other = pair_swap(mine);
    if constexpr (R == 32) return other;
    else if constexpr (R < 32) return __builtin_amdgcn_alignbit(mine, other, 32 - R);
    else return __builtin_amdgcn_alignbit(other, mine, 64 - R);
  }
}
__device__ __forceinline__ u32 x3(u32 a, u32 b, u32 c) { return (u32)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0x96); }
__device__ __forceinline__ u32 chi32(u32 a, u32 b, u32 c) { return a ^ (~b & c); }
#ifndef MZK_KECCAK_PAIR_ASM
#define MZK_KECCAK_PAIR_ASM 1       // 0: the compiler-scheduled round below (A/B builds)
#endif
__device__ __forceinline__ void keccak_f_pair(u32 (&a)[25], int parity) {
#if MZK_KECCAK_PAIR_ASM
  // One scheduled asm block per round (mzk_keccak_asm.h, generated): hipcc emitted the round word by word -- xor, s_nop 1,
  // v_mov_b32_dpp, s_nop 0, v_alignbit: 49 wait-state instructions per round, 7.6 cycles per instruction on the lone wave of a
  // tree's upper levels; batched (all xors, all lane exchanges, all funnel shifts) every hazard distance is covered by independent work.
  // the round constant of round r + 1 is fetched while round r runs: loaded at the top of the round it belongs to, the scalar load and
  // its wait (~100 cycles on the lone wave of a tree's upper levels, a fifth of the round) sat in front of every round
  // Where the round constant comes from (same-box A/B, profiles/round6_keccak_round_constant_ab.txt; FRI round at 2^14 / Merkle commit of 2^16 leaves):
  //   0  `KECCAK_RC[rnd]` inside its round: s_getpc + address arithmetic + s_load_dwordx2 + s_waitcnt in front of EVERY round of the lone wave
  //      of a tree's upper levels -- a fifth of the round                                                             114 - 116 us / 0.153 ms
  //   1  the constant of round r + 1 fetched while round r runs (the wait finds it there)                           106 - 108 us / 0.147 ms
  //   2  all 24 rounds unrolled, the constants literals of the instruction stream (49 KB of code in k_merkle_tail)  104 - 105 us / 0.139 ms
#ifndef MZK_KECCAK_RC_AHEAD
#define MZK_KECCAK_RC_AHEAD 2
#endif
#if MZK_KECCAK_RC_AHEAD == 2
  // all 24 rounds unrolled, the constants literals of the instruction stream: no scalar load, no address arithmetic per round
  constexpr u64 K[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
      0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
      0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll
  for (int rnd = 0; rnd < 24; rnd++) keccak_round_pair_asm(a, parity ? (u32)(K[rnd] >> 32) : (u32)K[rnd]);
#elif MZK_KECCAK_RC_AHEAD
  u64 rc = KECCAK_RC[0];
#pragma unroll 1
  for (int rnd = 0; rnd < 24; rnd++) {
    const u64 nxt = KECCAK_RC[(rnd + 1) & 31];          // (the table has 32 entries: the last fetch reads a pad word)
    keccak_round_pair_asm(a, parity ? (u32)(rc >> 32) : (u32)rc);
    rc = nxt;
  }
#else
#pragma unroll 1
  for (int rnd = 0; rnd < 24; rnd++) {
    const u64 rc = KECCAK_RC[rnd];
    keccak_round_pair_asm(a, parity ? (u32)(rc >> 32) : (u32)rc);
  }
#endif
  return;
#endif
#pragma unroll 1
  for (int rnd = 0; rnd < 24; rnd++) {
    const u32 c0 = x3(x3(a[0], a[5], a[10]), a[15], a[20]);
    const u32 c1 = x3(x3(a[1], a[6], a[11]), a[16], a[21]);
    const u32 c2 = x3(x3(a[2], a[7], a[12]), a[17], a[22]);
    const u32 c3 = x3(x3(a[3], a[8], a[13]), a[18], a[23]);
    const u32 c4 = x3(x3(a[4], a[9], a[14]), a[19], a[24]);
    const u32 d0 = c4 ^ pair_rot<1>(c1), d1 = c0 ^ pair_rot<1>(c2), d2 = c1 ^ pair_rot<1>(c3), d3 = c2 ^ pair_rot<1>(c4), d4 = c3 ^ pair_rot<1>(c0);
    const u32 b0 = pair_rot<0>(a[0] ^ d0), b16 = pair_rot<36>(a[5] ^ d0), b7 = pair_rot<3>(a[10] ^ d0), b23 = pair_rot<41>(a[15] ^ d0), b14 = pair_rot<18>(a[20] ^ d0);
    const u32 b10 = pair_rot<1>(a[1] ^ d1), b1 = pair_rot<44>(a[6] ^ d1), b17 = pair_rot<10>(a[11] ^ d1), b8 = pair_rot<45>(a[16] ^ d1), b24 = pair_rot<2>(a[21] ^ d1);
    const u32 b20 = pair_rot<62>(a[2] ^ d2), b11 = pair_rot<6>(a[7] ^ d2), b2 = pair_rot<43>(a[12] ^ d2), b18 = pair_rot<15>(a[17] ^ d2), b9 = pair_rot<61>(a[22] ^ d2);
    const u32 b5 = pair_rot<28>(a[3] ^ d3), b21 = pair_rot<55>(a[8] ^ d3), b12 = pair_rot<25>(a[13] ^ d3), b3 = pair_rot<21>(a[18] ^ d3), b19 = pair_rot<56>(a[23] ^ d3);
    const u32 b15 = pair_rot<27>(a[4] ^ d4), b6 = pair_rot<20>(a[9] ^ d4), b22 = pair_rot<39>(a[14] ^ d4), b13 = pair_rot<8>(a[19] ^ d4), b4 = pair_rot<14>(a[24] ^ d4);
    a[0] = chi32(b0, b1, b2); a[1] = chi32(b1, b2, b3); a[2] = chi32(b2, b3, b4); a[3] = chi32(b3, b4, b0); a[4] = chi32(b4, b0, b1);
    a[5] = chi32(b5, b6, b7); a[6] = chi32(b6, b7, b8); a[7] = chi32(b7, b8, b9); a[8] = chi32(b8, b9, b5); a[9] = chi32(b9, b5, b6);
    a[10] = chi32(b10, b11, b12); a[11] = chi32(b11, b12, b13); a[12] = chi32(b12, b13, b14); a[13] = chi32(b13, b14, b10); a[14] = chi32(b14, b10, b11);
    a[15] = chi32(b15, b16, b17); a[16] = chi32(b16, b17, b18); a[17] = chi32(b17, b18, b19); a[18] = chi32(b18, b19, b15); a[19] = chi32(b19, b15, b16);
    a[20] = chi32(b20, b21, b22); a[21] = chi32(b21, b22, b23); a[22] = chi32(b22, b23, b24); a[23] = chi32(b23, b24, b20); a[24] = chi32(b24, b20, b21);
    const u64 rc = KECCAK_RC[rnd];
    a[0] ^= parity ? (u32)(rc >> 32) : (u32)rc;
  }
}
// hash of two child digests by a lane pair: lane `parity` reads and writes its halves of the 64-bit words
__device__ __forceinline__ void sha3_of_two_digests_pair(const u64* __restrict__ children, u64* __restrict__ out, int parity) {
  u32 a[25];
  const u32* c32 = reinterpret_cast<const u32*>(children);
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = c32[2 * i + parity];
  a[8] = parity ? 0u : 0x06u;
#pragma unroll
  for (int i = 9; i < 25; i++) a[i] = 0;
  a[16] = parity ? 0x80000000u : 0u;
  keccak_f_pair(a, parity);
  u32* o32 = reinterpret_cast<u32*>(out);
#pragma unroll
  for (int i = 0; i < 4; i++) o32[2 * i + parity] = a[i];
}

constexpr int SHA3_RATE = 136;

// hash of one 64-byte message (two child digests): a[0..7] = data, pad 0x06 at byte 64, 0x80 at byte 135
__device__ __forceinline__ void sha3_of_two_digests(const u64* __restrict__ children, u64* __restrict__ out) {
  u64 a[25];
  const ulonglong2* c2 = reinterpret_cast<const ulonglong2*>(children);
#pragma unroll
  for (int i = 0; i < 4; i++) { ulonglong2 v = c2[i]; a[2 * i] = v.x; a[2 * i + 1] = v.y; }
  a[8] = 0x06ULL;
#pragma unroll
  for (int i = 9; i < 25; i++) a[i] = 0;
  a[16] = 0x8000000000000000ULL;
  keccak_f(a);
  ulonglong2* o2 = reinterpret_cast<ulonglong2*>(out);
  o2[0] = make_ulonglong2(a[0], a[1]);
  o2[1] = make_ulonglong2(a[2], a[3]);
}

// ---- level 1 from field elements -----------------------------------------------------------------------
// One lane per leaf pair.  The pair's message (<= 2 * (9 + 4 NW) <= 82 bytes) is serialised bytewise into a
// word-major LDS block buffer, padded, and absorbed as 17 lanes.
constexpr int LEAF_THREADS = 128;
template <int NW>
__global__ __launch_bounds__(LEAF_THREADS) void k_merkle_leaf_pairs(const u32* __restrict__ elems, size_t pairs, u64* __restrict__ nodes,
                                                                     const u8* __restrict__ neg) {
  __shared__ u32 blk[SHA3_RATE / 4][LEAF_THREADS];
  const int tid = threadIdx.x;
  const size_t i = (size_t)blockIdx.x * LEAF_THREADS + tid;
  if (i >= pairs) return;
#pragma unroll
  for (int w = 0; w < SHA3_RATE / 4; w++) blk[w][tid] = 0;
  auto put = [&](int pos, u32 byte) { reinterpret_cast<u8*>(&blk[pos >> 2][tid])[pos & 3] = (u8)byte; };
  int pos = 0;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    u32 w[NW];
    const uint4* p4 = reinterpret_cast<const uint4*>(elems + (2 * i + e) * NW);
#pragma unroll
    for (int q = 0; q < NW / 4; q++) { uint4 v = p4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    int k = 0;                                    // significant u32 digits (BigUint keeps no leading zero digit)
#pragma unroll
    for (int j = 0; j < NW; j++) if (w[j]) k = j + 1;
    // Sign as i8: Plus = 1, NoSign = 0, Minus = -1 (elements whose BigInt the reference left negative, field.rs:98-110;
    // `neg` is null for canonical codewords)
    put(pos, k ? ((neg && neg[2 * i + e]) ? 0xffu : 1u) : 0u);
    put(pos + 1, (u32)k);                          // sequence length as u64 LE (k <= 8: one non-zero byte)
    pos += 9;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      if (j < k) {
        put(pos, w[j] & 255u); put(pos + 1, (w[j] >> 8) & 255u); put(pos + 2, (w[j] >> 16) & 255u); put(pos + 3, w[j] >> 24);
        pos += 4;
      }
    }
  }
  put(pos, 0x06u);
  reinterpret_cast<u8*>(&blk[(SHA3_RATE - 1) >> 2][tid])[3] |= 0x80u;
  u64 a[25];
#pragma unroll
  for (int l = 0; l < SHA3_RATE / 8; l++) a[l] = (u64)blk[2 * l][tid] | ((u64)blk[2 * l + 1][tid] << 32);
#pragma unroll
  for (int l = SHA3_RATE / 8; l < 25; l++) a[l] = 0;
  keccak_f(a);
  ulonglong2* o2 = reinterpret_cast<ulonglong2*>(nodes + 4 * i);
  o2[0] = make_ulonglong2(a[0], a[1]);
  o2[1] = make_ulonglong2(a[2], a[3]);
}

// The same level by lane PAIRS, for trees whose leaf level is itself latency-bound (a FRI round's codeword: at most 2^15 pairs are one
// wave per SIMD at two lanes each): lane e of a pair serialises element e of its message -- the odd lane starts behind the even lane's
// element, whose digit count it recounts --, the halves of the 17 rate words come back from LDS and the permutation is the lane-pair one
// (~127 instructions per round and lane instead of 190).  Same bytes, same digests as k_merkle_leaf_pairs.
constexpr size_t LEAF_PAIR_MAX = (size_t)1 << 15;      // pairs
template <int NW>
__global__ __launch_bounds__(LEAF_THREADS) void k_merkle_leaf_pairs_lp(const u32* __restrict__ elems, size_t pairs, u64* __restrict__ nodes,
                                                                        const u8* __restrict__ neg) {
  __shared__ u32 blk[SHA3_RATE / 4][LEAF_THREADS / 2];
  const int tid = threadIdx.x, col = tid >> 1, parity = tid & 1;
  const size_t i = (size_t)blockIdx.x * (LEAF_THREADS / 2) + col;
  const bool live = i < pairs;                    // pair-uniform; every lane reaches the barriers
#pragma unroll
  for (int w = parity; w < SHA3_RATE / 4; w += 2) blk[w][col] = 0;
  __syncthreads();
  if (live) {
    auto put = [&](int pos, u32 byte) { atomicOr(&blk[pos >> 2][col], (byte & 255u) << (8 * (pos & 3))); };     // the two lanes may meet in one word
    u32 w0[NW], w[NW];
    {
      const uint4* p4 = reinterpret_cast<const uint4*>(elems + (2 * i) * NW);
#pragma unroll
      for (int q = 0; q < NW / 4; q++) { uint4 v = p4[q]; w0[4 * q] = v.x; w0[4 * q + 1] = v.y; w0[4 * q + 2] = v.z; w0[4 * q + 3] = v.w; }
    }
    int k0 = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) if (w0[j]) k0 = j + 1;
    if (parity) {
      const uint4* p4 = reinterpret_cast<const uint4*>(elems + (2 * i + 1) * NW);
#pragma unroll
      for (int q = 0; q < NW / 4; q++) { uint4 v = p4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    } else {
#pragma unroll
      for (int j = 0; j < NW; j++) w[j] = w0[j];
    }
    int k = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) if (w[j]) k = j + 1;
    int pos = parity ? 9 + 4 * k0 : 0;
    put(pos, k ? ((neg && neg[2 * i + parity]) ? 0xffu : 1u) : 0u);
    put(pos + 1, (u32)k);
    pos += 9;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      if (j < k) {
        put(pos, w[j] & 255u); put(pos + 1, (w[j] >> 8) & 255u); put(pos + 2, (w[j] >> 16) & 255u); put(pos + 3, w[j] >> 24);
        pos += 4;
      }
    }
    if (parity) {
      put(pos, 0x06u);
      put(SHA3_RATE - 1, 0x80u);
    }
  }
  __syncthreads();
  if (!live) return;
  u32 a[25];
#pragma unroll
  for (int l = 0; l < SHA3_RATE / 8; l++) a[l] = blk[2 * l + parity][col];
#pragma unroll
  for (int l = SHA3_RATE / 8; l < 25; l++) a[l] = 0;
  keccak_f_pair(a, parity);
  u32* o32 = reinterpret_cast<u32*>(nodes + 4 * i);
#pragma unroll
  for (int l = 0; l < 4; l++) o32[2 * l + parity] = a[l];
}

// ---- level 1 from arbitrary byte leaves: leaf 2i || leaf 2i+1 is the contiguous range off[2i] .. off[2i+2) ---
__global__ __launch_bounds__(128) void k_merkle_leaf_pairs_bytes(const u8* __restrict__ leaves, const u64* __restrict__ off, size_t pairs,
                                                                 u64* __restrict__ nodes) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pairs) return;
  const u8* msg = leaves + off[2 * i];
  const size_t len = (size_t)(off[2 * i + 2] - off[2 * i]);
  u64 a[25];
#pragma unroll
  for (int l = 0; l < 25; l++) a[l] = 0;
  size_t base = 0;
  for (;;) {
    const bool last = (len - base) < (size_t)SHA3_RATE;      // the padded final block
#pragma unroll
    for (int l = 0; l < SHA3_RATE / 8; l++) {
      u64 wv = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const size_t idx = base + 8 * l + t;
        u64 b = idx < len ? (u64)msg[idx] : (idx == len ? 0x06ULL : 0ULL);
        wv |= b << (8 * t);
      }
      a[l] ^= wv;
    }
    if (last) a[SHA3_RATE / 8 - 1] ^= 0x8000000000000000ULL;
    keccak_f(a);
    if (last) break;
    base += SHA3_RATE;
  }
  ulonglong2* o2 = reinterpret_cast<ulonglong2*>(nodes + 4 * i);
  o2[0] = make_ulonglong2(a[0], a[1]);
  o2[1] = make_ulonglong2(a[2], a[3]);
}

// (magnitude, Sign::Minus) -> canonical representative p - magnitude, in place (magnitude 0 stays 0: BigInt has no -0)
template <int NW>
__global__ __launch_bounds__(256) void k_canonicalize_signed(u32* __restrict__ elems, const u8* __restrict__ neg, size_t n, int fid) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !neg[i]) return;
  u32 w[NW], any = 0;
#pragma unroll
  for (int j = 0; j < NW; j++) { w[j] = elems[i * NW + j]; any |= w[j]; }
  if (!any) return;
  u32 pw[8];
  if (NW == 4) { Fe<M128Params> pm; for (int k = 0; k < M128Params::L; k++) pm.l[k] = M128Params::P[k]; fe_pack<M128Params>(pm, pw); }
  else { Fe<FrParams> pm; for (int k = 0; k < FrParams::L; k++) pm.l[k] = FrParams::P[k]; fe_pack<FrParams>(pm, pw); }
  u32 borrow = 0;
#pragma unroll
  for (int j = 0; j < NW; j++) {
    const u64 d = (u64)pw[j] - w[j] - borrow;
    elems[i * NW + j] = (u32)d;
    borrow = (u32)(d >> 32) & 1u;
  }
}

// ---- ragged leaf counts (merkle.rs:15-25 accepts any non-empty slice: mid = len / 2) ----------------------------
// The recursion splits k leaves into floor(k/2) | ceil(k/2), so at depth D = floor(log2 n) there are M = 2^D subtrees
// of one or two leaves.  Define item p = the commitment of subtree p: the leaf ITSELF (one leaf, merkle.rs:17-19) or
// hash(leaf || leaf).  Everything above depth D is a full binary tree, hence
//     Merkle::commit(n ragged leaves) == Merkle::commit(M power-of-two items),
// and the kernel below only has to produce the items (bytes, at host-computed offsets); the power-of-two byte-leaf
// path does the rest.  node_start[p] = first leaf of subtree p (node_start[M] = n).
__global__ __launch_bounds__(128) void k_merkle_ragged_items(const u8* __restrict__ leaves, const u64* __restrict__ off, const u32* __restrict__ node_start,
                                                             size_t m, const u64* __restrict__ item_off, u8* __restrict__ items) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const u32 first = node_start[p], cnt = node_start[p + 1] - first;
  const u8* msg = leaves + off[first];
  const size_t len = (size_t)(off[first + cnt] - off[first]);
  u8* dst = items + item_off[p];
  if (cnt == 1) {
    for (size_t i = 0; i < len; i++) dst[i] = msg[i];
    return;
  }
  u64 a[25];
#pragma unroll
  for (int l = 0; l < 25; l++) a[l] = 0;
  size_t base = 0;
  for (;;) {
    const bool last = (len - base) < (size_t)SHA3_RATE;
#pragma unroll
    for (int l = 0; l < SHA3_RATE / 8; l++) {
      u64 wv = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const size_t idx = base + 8 * l + t;
        u64 b = idx < len ? (u64)msg[idx] : (idx == len ? 0x06ULL : 0ULL);
        wv |= b << (8 * t);
      }
      a[l] ^= wv;
    }
    if (last) a[SHA3_RATE / 8 - 1] ^= 0x8000000000000000ULL;
    keccak_f(a);
    if (last) break;
    base += SHA3_RATE;
  }
  for (int q = 0; q < 4; q++)
    for (int t = 0; t < 8; t++) dst[8 * q + t] = (u8)(a[q] >> (8 * t));
}

// ---- inner levels -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_merkle_level(const u64* __restrict__ below, size_t count, u64* __restrict__ above) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  sha3_of_two_digests(below + 8 * i, above + 4 * i);
}
// small levels (latency-bound: fewer hashes than the GPU has lanes to spare): one lane PAIR per hash
constexpr size_t LEVEL_PAIR_MAX = 16384;      // hashes
__global__ __launch_bounds__(128) void k_merkle_level_pair(const u64* __restrict__ below, size_t count, u64* __restrict__ above) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;
  if (i >= count) return;                    // pair-uniform
  sha3_of_two_digests_pair(below + 8 * i, above + 4 * i, (int)(t & 1));
}
// LV consecutive small levels in one launch (round 5): a workgroup hashes 32 << LV parents from global children, then their parents from
// the digests it just left in LDS, and so on -- each level after the first costs one hash (~5 us) instead of a launch of its own (8.7 us:
// the same hash plus the dependent kernel boundary).  Every digest still goes to global memory for the authentication paths.
template <int LV>
__global__ __launch_bounds__(64 << LV) void k_merkle_level_pair_multi(const u64* __restrict__ below, size_t parents, u64* __restrict__ above) {
  constexpr int H0 = 32 << LV;
  __shared__ u32 sh[2][H0 * 8];
  const int h = threadIdx.x >> 1, parity = threadIdx.x & 1;
  size_t cnt = parents, base = (size_t)blockIdx.x * H0;
  u64* dst = above;
  int width = H0;
#pragma unroll 1
  for (int lv = 0; lv < LV; lv++) {
    if (h < width && base + h < cnt) {                 // pair-uniform
      u32 a[25];
      const u32* c32 = lv == 0 ? reinterpret_cast<const u32*>(below + 8 * (base + h)) : sh[(lv - 1) & 1] + 16 * h;
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = c32[2 * i + parity];
      a[8] = parity ? 0u : 0x06u;
#pragma unroll
      for (int i = 9; i < 25; i++) a[i] = 0;
      a[16] = parity ? 0x80000000u : 0u;
      keccak_f_pair(a, parity);
      u32* o32 = reinterpret_cast<u32*>(dst + 4 * (base + h));
      u32* l32 = sh[lv & 1] + 8 * h;
#pragma unroll
      for (int i = 0; i < 4; i++) { o32[2 * i + parity] = a[i]; l32[2 * i + parity] = a[i]; }
    }
    __syncthreads();
    dst += 4 * cnt;
    cnt >>= 1; base >>= 1; width >>= 1;
  }
}
// the last levels (<= TAIL_NODES nodes each) in one workgroup: no launch per level; one lane pair per hash
constexpr int TAIL_NODES = 512;
__global__ __launch_bounds__(TAIL_NODES) void k_merkle_tail(u64* __restrict__ level, size_t count, size_t stop, u32* __restrict__ mailbox, u32 seq) {
  // `level` holds `count` nodes; the levels above follow contiguously (count/2, count/4, ... stop); stop = 1 for one tree,
  // the number of trees for a batch (their roots are the last level).
  // The digests travel from level to level through LDS (ping-pong): every level is one dependent hash on an almost empty CU, and
  // reading the children back from global memory behind a fence was a ~1.5-us round trip per level on top of the ~7 us of the hash.
  // Global memory still receives every digest (the authentication paths are gathered from there) but nobody waits for it.
  __shared__ u32 sh[2][TAIL_NODES * 8];
  const int tid = threadIdx.x;
  {
    const u32* g = reinterpret_cast<const u32*>(level);
    for (size_t i = tid; i < count * 8; i += TAIL_NODES) sh[0][i] = g[i];
  }
  __syncthreads();
  u64* below = level;
  int cur = 0;
  while (count > stop) {
    const size_t up = count / 2;
    u64* above = below + 4 * count;
    const int h = tid >> 1, parity = tid & 1;
    if ((size_t)h < up) {
      u32 a[25];
      const u32* c32 = sh[cur] + 16 * h;
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = c32[2 * i + parity];
      a[8] = parity ? 0u : 0x06u;
#pragma unroll
      for (int i = 9; i < 25; i++) a[i] = 0;
      a[16] = parity ? 0x80000000u : 0u;
      keccak_f_pair(a, parity);
      u32* o32 = reinterpret_cast<u32*>(above + 4 * h);
      u32* l32 = sh[cur ^ 1] + 8 * h;
#pragma unroll
      for (int i = 0; i < 4; i++) { o32[2 * i + parity] = a[i]; l32[2 * i + parity] = a[i]; }
    }
    __syncthreads();
    below = above;
    count = up;
    cur ^= 1;
  }
  // one tree whose root the host is waiting for (a FRI round's transcript, Merkle::commit): the root goes straight into the caller's mapped
  // host buffer, then a fence, then the sequence number the host spins on -- no copy engine, no stream synchronize (merkle_root_to_host)
  if (mailbox != nullptr && tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) __builtin_nontemporal_store(sh[cur][i], mailbox + i);
    __threadfence_system();
    __hip_atomic_store(mailbox + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// authentication path: node (index >> l) ^ 1 of level l, l = 1 .. depth-1; level l starts at node n - n / 2^(l-1)
__global__ void k_merkle_gather(const u64* __restrict__ nodes, size_t n, size_t index, int depth, u64* __restrict__ out) {
  const int l = 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= depth) return;
  const size_t start = n - (n >> (l - 1));
  const u64* src = nodes + 4 * (start + ((index >> l) ^ 1));
#pragma unroll
  for (int q = 0; q < 4; q++) out[4 * (l - 1) + q] = src[q];
}

// many authentication paths at once (the FRI query phase opens three indices per colinearity test and round): block q
// gathers the digests of path q and the limbs of its sibling leaf
__global__ __launch_bounds__(64) void k_merkle_gather_batch(const u64* __restrict__ nodes, const u32* __restrict__ leaves, int leaf_words, size_t n,
                                                            const u64* __restrict__ indices, int depth, u64* __restrict__ out_nodes,
                                                            u32* __restrict__ out_leaves) {
  const size_t q = blockIdx.x;
  const size_t index = indices[q];
  const int l = 1 + threadIdx.x;
  if (l < depth) {
    const size_t start = n - (n >> (l - 1));
    const u64* src = nodes + 4 * (start + ((index >> l) ^ 1));
#pragma unroll
    for (int k = 0; k < 4; k++) out_nodes[(q * (size_t)(depth - 1) + (l - 1)) * 4 + k] = src[k];
  }
  if ((int)threadIdx.x < leaf_words) out_leaves[q * leaf_words + threadIdx.x] = leaves[(index ^ 1) * leaf_words + threadIdx.x];
}

// out[q] = leaves[indices[q]] (whole elements of leaf_words 32-bit words)
__global__ void k_gather_leaves(const u32* __restrict__ leaves, int leaf_words, const u64* __restrict__ indices, size_t count, u32* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count * (size_t)leaf_words) return;
  const size_t q = i / (size_t)leaf_words, w = i - q * (size_t)leaf_words;
  out[i] = leaves[indices[q] * (size_t)leaf_words + w];
}

}  // namespace mzk

using namespace mzk;

struct mzk_merkle {
  int kind;              // 0 = field elements, 1 = byte leaves
  int field;
  size_t n;
  int depth;             // log2 n
  u64* d_nodes;          // n - 1 digests: level 1 (n/2), level 2 (n/4), ..., root
  void* d_leaves;        // owned copy of the elements / the leaf bytes (needed by open and by n == 1)
  std::vector<uint64_t> offsets;   // byte leaves only
  std::vector<uint8_t> neg;        // field leaves given as (magnitude, sign): 1 = Sign::Minus; empty = all non-negative
  hipStream_t stream;    // the stream the tree was BUILT on (a caller stream for the *_dev builders); never used again
  // Owner: the context (and its device) whose memory holds the tree.  Every later operation enters that context, runs on
  // ITS stream and workspace (MerkleScope) behind `built`, whatever context is current and whatever became of `stream`.
  int ctx_index = 0;
  int device = -1;
  hipEvent_t built = nullptr;    // recorded behind the last kernel of the build
  // ragged leaf counts: the tree is built over m = 2^floor(log2 n) items (see k_merkle_ragged_items); d_nodes, depth
  // and the gather kernel then refer to the item tree
  // FRI::commit with the trees kept: the codewords and digests of ALL rounds live in one device allocation shared by the rounds'
  // handles (one hipMalloc per commit instead of two per round, and no copies: a round hashes and folds in place there);
  // d_leaves / d_nodes then point into it and the last handle freed releases it.
  struct SharedBlock { void* p; int refs; };
  SharedBlock* shared = nullptr;
  bool owns_leaves = false;        // with `shared`: d_leaves is nevertheless an allocation of its own (signed round-0 leaves)
  bool ragged = false;
  size_t m = 0;
  void* d_items = nullptr;
  std::vector<uint64_t> item_off;
  std::vector<uint32_t> node_start;
};

namespace mzk {
static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }

// end of a successful build on stream s of the current context
// complete_on_return: the caller synchronizes the stream before it hands the tree out (mzk_fri_commit does once, after its last round --
// since round 6 a round's root reaches the transcript through the mailbox without a synchronize; an error return frees the trees, and
// hipFree waits for the device), so there is nothing to order later calls behind: no event (one marker packet per round less)
static int merkle_stamp(mzk_merkle* t, hipStream_t s, bool complete_on_return = false) {
  t->ctx_index = ctx().index;
  t->device = ctx().device;
  if (complete_on_return) return MZK_OK;
  if (!t->built) MZK_HIP(hipEventCreateWithFlags(&t->built, hipEventDisableTiming));
  MZK_HIP(hipEventRecord(t->built, s));
  return MZK_OK;
}
// Enter the owning context of a tree for the scope of one entry point.  rc != MZK_OK: the context is gone or drives another
// device now (mzk_init_devices since the build) -- MZK_E_ARG, as srs_check_ctx reports a foreign SRS handle.
struct MerkleScope {
  CtxScope sc;
  int rc;
  hipStream_t s;
  explicit MerkleScope(const mzk_merkle* t) : sc(t->ctx_index), rc(MZK_OK), s(nullptr) {
    if (!sc.ok || ctx().device != t->device) {
      set_error("Merkle tree handle was built on context %d (device %d); that context %s", t->ctx_index, t->device,
                sc.ok ? "drives another device now" : "no longer exists");
      rc = MZK_E_ARG;
      return;
    }
    s = ctx().stream;
    if (t->built && hipStreamWaitEvent(s, t->built, 0) != hipSuccess) { (void)hipGetLastError(); rc = MZK_E_HIP; set_error("merkle: cannot order behind the build"); }
  }
};

// hashes level 1 .. root into d_nodes; d_leaves / d_off already on the device
// `trees` > 1: n = trees * (leaves per tree), all trees of one power-of-two size, leaves back to back.  Level l of the whole
// array is then level l of every tree side by side (pairs never straddle trees), so a batch is the bottom of one big tree,
// hashed down to `trees` nodes: the roots, at d_nodes + 4 * (n - 2 * trees).
// mailbox / seq: see merkle_root_to_host; *mailed = the tail ran and will post the root there
static int merkle_hash_levels(int kind, int fid, const void* d_leaves, const u64* d_off, size_t n, u64* d_nodes, hipStream_t s,
                              const u8* d_neg = nullptr, size_t trees = 1, u32* mailbox = nullptr, u32 seq = 0, bool* mailed = nullptr) {
  if (mailed) *mailed = false;
  if (n < 2) return MZK_OK;
  ProfScope ps(s, MZK_PH_MERKLE);
  const size_t pairs = n / 2;
  const unsigned blocks = (unsigned)((pairs + 127) / 128);
  if (kind == 1)
    hipLaunchKernelGGL(k_merkle_leaf_pairs_bytes, dim3(blocks), dim3(128), 0, s, (const u8*)d_leaves, d_off, pairs, d_nodes);
  else {
    static const int lp_on = tune_int("MZK_LEAF_LANE_PAIRS", 1);      // tuning build: 0 = one lane per leaf pair at every size (A/B)
    const bool lp = lp_on && pairs <= LEAF_PAIR_MAX;
    const unsigned lpb = (unsigned)((pairs + LEAF_THREADS / 2 - 1) / (LEAF_THREADS / 2));
    if (fid == MZK_FIELD_M128) {
      if (lp) hipLaunchKernelGGL((k_merkle_leaf_pairs_lp<4>), dim3(lpb), dim3(LEAF_THREADS), 0, s, (const u32*)d_leaves, pairs, d_nodes, d_neg);
      else hipLaunchKernelGGL((k_merkle_leaf_pairs<4>), dim3(blocks), dim3(LEAF_THREADS), 0, s, (const u32*)d_leaves, pairs, d_nodes, d_neg);
    } else {
      if (lp) hipLaunchKernelGGL((k_merkle_leaf_pairs_lp<8>), dim3(lpb), dim3(LEAF_THREADS), 0, s, (const u32*)d_leaves, pairs, d_nodes, d_neg);
      else hipLaunchKernelGGL((k_merkle_leaf_pairs<8>), dim3(blocks), dim3(LEAF_THREADS), 0, s, (const u32*)d_leaves, pairs, d_nodes, d_neg);
    }
  }
  u64* below = d_nodes;
  size_t count = pairs;
  while (count > (size_t)TAIL_NODES && count > trees) {
    u64* above = below + 4 * count;
    if (count / 2 <= LEVEL_PAIR_MAX && (count >> 3) >= (size_t)TAIL_NODES && (count >> 3) >= trees) {          // three levels in one launch
      hipLaunchKernelGGL((k_merkle_level_pair_multi<3>), dim3((unsigned)((count / 2 + 255) / 256)), dim3(512), 0, s, (const u64*)below, count / 2, above);
      below = above + 4 * (count / 2) + 4 * (count / 4);
      count >>= 3;
      continue;
    }
    if (count / 2 <= LEVEL_PAIR_MAX && (count >> 2) >= (size_t)TAIL_NODES && (count >> 2) >= trees) {          // two
      hipLaunchKernelGGL((k_merkle_level_pair_multi<2>), dim3((unsigned)((count / 2 + 127) / 128)), dim3(256), 0, s, (const u64*)below, count / 2, above);
      below = above + 4 * (count / 2);
      count >>= 2;
      continue;
    }
    if (count / 2 <= LEVEL_PAIR_MAX)
      hipLaunchKernelGGL(k_merkle_level_pair, dim3((unsigned)((count + 127) / 128)), dim3(128), 0, s, (const u64*)below, count / 2, above);
    else
      hipLaunchKernelGGL(k_merkle_level, dim3((unsigned)((count / 2 + 127) / 128)), dim3(128), 0, s, (const u64*)below, count / 2, above);
    below = above;
    count /= 2;
  }
  if (count > trees) {
    const bool mail = mailbox != nullptr && trees == 1;
    hipLaunchKernelGGL(k_merkle_tail, dim3(1), dim3(TAIL_NODES), 0, s, below, count, trees, mail ? mailbox : (u32*)nullptr, seq);
    if (mail && mailed) *mailed = true;
  }
  MZK_HIP(hipGetLastError());
  return MZK_OK;
}
// The root of ONE tree to the host.  The copy engine and the stream synchronize behind it cost ~15 us per FRI round; the tail kernel instead
// writes the 32 bytes and a sequence number into the context's pinned landing zone (mapped into the device's address space) and the host spins on
// the number.  Stream errors still surface: every 1024 spins the stream is queried, and when it has drained without the number showing up
// (or the tail was not part of this tree: two leaves) the root is fetched the ordinary way.  The stream is NOT synchronized on return.
struct RootMailbox { u32* p; u32 seq; };
static u32 g_mailbox_seq[MZK_MAX_CTX];        // per context: the number of the last root a tail kernel was asked to post
static int merkle_mailbox(RootMailbox* mb) {
  Context& c = ctx();
  if (!c.bounce) MZK_HIP(hipHostMalloc(&c.bounce, SMALL_D2H_MAX, hipHostMallocPortable));
  static const int enabled = tune_int("MZK_ROOT_MAILBOX", 1);     // tuning build: 0 = copy + synchronize (A/B)
  mb->p = enabled ? (u32*)c.bounce + 512 : nullptr;                 // the second half of the landing zone: d2h_sync uses the first bytes
  mb->seq = ++g_mailbox_seq[c.index];
  if (mb->seq == 0) mb->seq = ++g_mailbox_seq[c.index];
  if (mb->p) ((volatile u32*)mb->p)[8] = 0;                         // (a larger d2h_sync may have run over this half since the last root)
  return MZK_OK;
}
static int merkle_root_to_host(uint8_t* root, const u64* d_root, const RootMailbox& mb, bool mailed, hipStream_t s) {
  if (mailed && mb.p) {
    volatile u32* flag = mb.p + 8;
    for (unsigned spins = 1;; spins++) {
      if (*flag == mb.seq) { std::atomic_thread_fence(std::memory_order_acquire); memcpy(root, (const void*)mb.p, 32); return MZK_OK; }
      if ((spins & 1023) == 0) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) { if (*flag == mb.seq) continue; break; }       // drained: the number is there or never comes
        if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
      }
    }
  }
  return d2h_sync(root, d_root, 32, s);
}

// bincode(FiniteFieldElement) of ONE canonical element on the host: used for the n == 1 root and for the
// sibling leaf of an authentication path (both are a copy of input bytes, not computation).
static size_t host_bincode_field(const uint64_t* limbs, int nl, uint8_t* out, bool negative = false) {
  int k = 2 * nl;
  while (k > 0 && (uint32_t)(limbs[(k - 1) / 2] >> (32 * ((k - 1) & 1))) == 0) k--;
  out[0] = k ? (negative ? 0xff : 1) : 0;
  uint64_t len = (uint64_t)k;
  memcpy(out + 1, &len, 8);
  for (int i = 0; i < k; i++) { uint32_t d = (uint32_t)(limbs[i / 2] >> (32 * (i & 1))); memcpy(out + 9 + 4 * i, &d, 4); }
  return 9 + 4 * (size_t)k;
}

static int merkle_build(int kind, int fid, const void* src, bool src_on_device, size_t leaf_bytes, const uint64_t* offsets, size_t n,
                        mzk_merkle** out, hipStream_t s, const uint8_t* neg_host = nullptr);

// Any leaf count that is not a power of two (never produced by the provers, but accepted by merkle.rs:15-25).
static int merkle_build_ragged(const uint8_t* leaves, const uint64_t* offsets, size_t n, mzk_merkle** out, hipStream_t s) {
  if (n > ((size_t)1 << 31)) { set_error("merkle: ragged trees support up to 2^31 leaves"); return MZK_E_ARG; }
  mzk_merkle* t = new mzk_merkle();
  t->kind = 1; t->field = -1; t->n = n; t->stream = s; t->d_nodes = nullptr; t->d_leaves = nullptr; t->ragged = true;
  int D = 0;
  while (((size_t)2 << D) <= n) D++;
  const size_t m = (size_t)1 << D;
  t->m = m; t->depth = D;
  t->offsets.assign(offsets, offsets + n + 1);
  // subtree p at depth D: walk the bits of p from the root (0 = left = floor half, 1 = right = ceil half)
  t->node_start.resize(m + 1);
  t->item_off.resize(m + 1);
  for (size_t p = 0; p < m; p++) {
    size_t start = 0, size = n;
    for (int b = D - 1; b >= 0; b--) {
      const size_t l = size / 2;
      if ((p >> b) & 1) { start += l; size -= l; } else size = l;
    }
    t->node_start[p] = (uint32_t)start;
  }
  t->node_start[m] = (uint32_t)n;
  uint64_t io = 0;
  for (size_t p = 0; p < m; p++) {
    t->item_off[p] = io;
    const uint32_t first = t->node_start[p], cnt = t->node_start[p + 1] - first;
    io += cnt == 1 ? offsets[first + 1] - offsets[first] : 32;
  }
  t->item_off[m] = io;
  const size_t leaf_bytes = (size_t)(offsets[n] - offsets[0]);
  u64 *d_off = nullptr, *d_ioff = nullptr;
  u32* d_ns = nullptr;
  int rc = MZK_OK;
  do {
    if (hipMalloc(&t->d_leaves, leaf_bytes ? leaf_bytes : 16) != hipSuccess || hipMalloc(&t->d_items, io ? io : 16) != hipSuccess ||
        hipMalloc((void**)&t->d_nodes, (m - 1) * 32) != hipSuccess) { set_error("merkle: hipMalloc failed"); rc = MZK_E_HIP; break; }
    if ((rc = ws_get(WS_MISC_D, (n + 1) * 8, (void**)&d_off)) != MZK_OK) break;
    if ((rc = ws_get(WS_MISC_E, (m + 1) * 8, (void**)&d_ioff)) != MZK_OK) break;
    if ((rc = ws_get(WS_MISC_F, (m + 1) * 4, (void**)&d_ns)) != MZK_OK) break;
    if ((leaf_bytes && hipMemcpyAsync(t->d_leaves, leaves, leaf_bytes, hipMemcpyHostToDevice, s) != hipSuccess) ||
        hipMemcpyAsync(d_off, offsets, (n + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_ioff, t->item_off.data(), (m + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_ns, t->node_start.data(), (m + 1) * 4, hipMemcpyHostToDevice, s) != hipSuccess) { set_error("merkle: copy failed"); rc = MZK_E_HIP; break; }
    hipLaunchKernelGGL(k_merkle_ragged_items, dim3((unsigned)((m + 127) / 128)), dim3(128), 0, s, (const u8*)t->d_leaves, (const u64*)d_off, (const u32*)d_ns, m,
                       (const u64*)d_ioff, (u8*)t->d_items);
    rc = merkle_hash_levels(1, -1, t->d_items, d_ioff, m, t->d_nodes, s);
    if (rc == MZK_OK && hipStreamSynchronize(s) != hipSuccess) { set_error("merkle: sync failed"); rc = MZK_E_HIP; }
    if (rc == MZK_OK) rc = merkle_stamp(t, s);
  } while (0);
  if (rc != MZK_OK) { mzk_merkle_free(t); return rc; }
  *out = t;
  return MZK_OK;
}

static int merkle_build(int kind, int fid, const void* src, bool src_on_device, size_t leaf_bytes, const uint64_t* offsets, size_t n,
                        mzk_merkle** out, hipStream_t s, const uint8_t* neg_host) {
  if (!out) { set_error("merkle: null output handle"); return MZK_E_ARG; }
  *out = nullptr;
  if (n == 0) { set_error("merkle: empty leaf set (Merkle::commit recurses forever on it, merkle.rs:20-22)"); return MZK_E_LENGTH; }
  if (!src) { set_error("merkle: null pointer"); return MZK_E_ARG; }
  if (!is_pow2(n)) {
    if (kind == 1) return merkle_build_ragged((const uint8_t*)src, offsets, n, out, s);
    // field elements: serialise bincode(FiniteFieldElement) on the host and take the byte-leaf path (not a prover path:
    // every codeword the reference commits to has a power-of-two length)
    const int nl = field_limbs64(fid);
    std::vector<uint64_t> host(n * (size_t)nl);
    if (src_on_device) {
      MZK_TRY(d2h_sync(host.data(), src, leaf_bytes, s));
    } else {
      memcpy(host.data(), src, leaf_bytes);
    }
    std::vector<uint8_t> blob(n * (size_t)(9 + 8 * nl));
    std::vector<uint64_t> off(n + 1);
    uint64_t o = 0;
    for (size_t i = 0; i < n; i++) { off[i] = o; o += host_bincode_field(host.data() + i * nl, nl, blob.data() + o, neg_host && neg_host[i]); }
    off[n] = o;
    return merkle_build_ragged(blob.data(), off.data(), n, out, s);
  }
  mzk_merkle* t = new mzk_merkle();
  t->kind = kind; t->field = fid; t->n = n; t->stream = s; t->d_nodes = nullptr; t->d_leaves = nullptr;
  t->depth = 0;
  while (((size_t)1 << t->depth) < n) t->depth++;
  u64* d_off = nullptr;
  int rc = MZK_OK;
  do {
    if (hipMalloc(&t->d_leaves, leaf_bytes ? leaf_bytes : 16) != hipSuccess) { set_error("merkle: hipMalloc failed"); rc = MZK_E_HIP; break; }
    if (leaf_bytes && hipMemcpyAsync(t->d_leaves, src, leaf_bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s) != hipSuccess) {
      set_error("merkle: leaf copy failed"); rc = MZK_E_HIP; break;
    }
    if (n > 1 && hipMalloc((void**)&t->d_nodes, (n - 1) * 32) != hipSuccess) { set_error("merkle: hipMalloc failed"); rc = MZK_E_HIP; break; }
    if (kind == 1) {
      t->offsets.assign(offsets, offsets + n + 1);
      if ((rc = ws_get(WS_MISC_D, (n + 1) * 8, (void**)&d_off)) != MZK_OK) break;
      if (hipMemcpyAsync(d_off, offsets, (n + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess) { set_error("merkle: offset copy failed"); rc = MZK_E_HIP; break; }
    }
    u8* d_neg = nullptr;
    if (kind == 0 && neg_host) {
      t->neg.assign(neg_host, neg_host + n);
      if ((rc = ws_get(WS_MISC_E, n, (void**)&d_neg)) != MZK_OK) break;
      if (hipMemcpyAsync(d_neg, neg_host, n, hipMemcpyHostToDevice, s) != hipSuccess) { set_error("merkle: sign copy failed"); rc = MZK_E_HIP; break; }
    }
    rc = merkle_hash_levels(kind, fid, t->d_leaves, d_off, n, t->d_nodes, s, d_neg);
    if (rc == MZK_OK && kind == 1 && hipStreamSynchronize(s) != hipSuccess) { set_error("merkle: sync failed"); rc = MZK_E_HIP; }
    if (rc == MZK_OK) rc = merkle_stamp(t, s);
  } while (0);
  if (rc != MZK_OK) { mzk_merkle_free(t); return rc; }
  *out = t;
  return MZK_OK;
}
}  // namespace mzk

extern "C" {

int mzk_merkle_build_field_dev(int field_id, const void* d_elems, size_t n, mzk_merkle** out, void* stream) {
  MZK_ENTER();
  WsGuard wsg((hipStream_t)stream);
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("merkle: bad field id %d", field_id); return MZK_E_ARG; }
  return merkle_build(0, field_id, d_elems, true, n * field_bytes(field_id), nullptr, n, out, (hipStream_t)stream);
}
static int build_field_host(int field_id, const uint64_t* elems, const uint8_t* negative, size_t n, mzk_merkle** out) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("merkle: bad field id %d", field_id); return MZK_E_ARG; }
  if (elems) {
    const HostField* hf = host_field(field_id);
    for (size_t i = 0; i < n; i++)
      if (!h_is_canonical(hf, elems + (size_t)hf->nl * i)) { set_error("merkle: element %zu not canonical", i); return MZK_E_RANGE; }
  }
  WsGuard wsg(ctx().stream);
  MZK_TRY(merkle_build(0, field_id, elems, false, n * field_bytes(field_id), nullptr, n, out, ctx().stream, negative));
  MZK_HIP(hipStreamSynchronize(ctx().stream));
  return MZK_OK;
}
int mzk_merkle_build_field(int field_id, const uint64_t* elems, size_t n, mzk_merkle** out) { return build_field_host(field_id, elems, nullptr, n, out); }
int mzk_merkle_build_field_signed(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, mzk_merkle** out) {
  return build_field_host(field_id, magnitudes, negative, n, out);
}
int mzk_merkle_commit_field_signed(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, uint8_t* root, size_t cap,
                                   size_t* root_len) {
  mzk_merkle* t = nullptr;
  MZK_TRY(build_field_host(field_id, magnitudes, negative, n, &t));
  const int rc = mzk_merkle_root(t, root, cap, root_len);
  mzk_merkle_free(t);
  return rc;
}
int mzk_merkle_build_bytes(const uint8_t* leaves, const uint64_t* offsets, size_t n, mzk_merkle** out) {
  MZK_ENTER();
  if (!offsets) { set_error("merkle: null offsets"); return MZK_E_ARG; }
  for (size_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) { set_error("merkle: offsets must be non-decreasing"); return MZK_E_ARG; }
  const size_t total = n ? (size_t)(offsets[n] - offsets[0]) : 0;
  uint8_t dummy = 0;
  if (!leaves) { if (total) { set_error("merkle: null pointer"); return MZK_E_ARG; } leaves = &dummy; }
  // offsets are rebased to the copied range
  std::vector<uint64_t> off(n + 1);
  for (size_t i = 0; i <= n; i++) off[i] = offsets[i] - offsets[0];
  WsGuard wsg(ctx().stream);
  return merkle_build(1, -1, leaves + offsets[0], false, total, off.data(), n, out, ctx().stream);
}

int mzk_merkle_root(const mzk_merkle* t, uint8_t* root, size_t cap, size_t* root_len) {
  if (!t || !root || !root_len) { set_error("merkle_root: null pointer"); return MZK_E_ARG; }
  MZK_ENTER();
  MerkleScope ms(t);
  MZK_TRY(ms.rc);
  if (t->n == 1) {   // merkle.rs:17-19: the single leaf itself
    uint8_t buf[48];
    size_t len;
    if (t->kind == 0) {
      uint64_t limbs[4];
      MZK_TRY(d2h_sync(limbs, t->d_leaves, field_bytes(t->field), ms.s));
      len = host_bincode_field(limbs, field_limbs64(t->field), buf, !t->neg.empty() && t->neg[0]);
      if (cap < len) { set_error("merkle_root: buffer too small (%zu < %zu)", cap, len); return MZK_E_LENGTH; }
      memcpy(root, buf, len);
    } else {
      len = (size_t)(t->offsets[1] - t->offsets[0]);
      if (cap < len) { set_error("merkle_root: buffer too small (%zu < %zu)", cap, len); return MZK_E_LENGTH; }
      MZK_TRY(d2h_sync(root, t->d_leaves, len, ms.s));
    }
    *root_len = len;
    return MZK_OK;
  }
  if (cap < 32) { set_error("merkle_root: buffer too small"); return MZK_E_LENGTH; }
  MZK_TRY(d2h_sync(root, t->d_nodes + 4 * ((t->ragged ? t->m : t->n) - 2), 32, ms.s));
  *root_len = 32;
  return MZK_OK;
}

int mzk_merkle_open_batch(const mzk_merkle* t, const uint64_t* indices, size_t count, uint8_t* paths, size_t stride, uint64_t* path_lens,
                          size_t* depth);
int mzk_merkle_open(const mzk_merkle* t, size_t index, uint8_t* path, size_t stride, uint64_t* path_len, size_t* depth) {
  if (!t || !path || !path_len || !depth) { set_error("merkle_open: null pointer"); return MZK_E_ARG; }
  if (t->n < 2) { set_error("merkle_open: needs at least two leaves (merkle.rs:32)"); return MZK_E_LENGTH; }
  if (index >= t->n) { set_error("merkle_open: index %zu out of range", index); return MZK_E_LENGTH; }
  if (stride < 32) { set_error("merkle_open: stride < 32"); return MZK_E_LENGTH; }
  if (t->kind == 0 && !t->ragged) {        // field-element tree: sibling leaf and digests in one gather and one synchronisation
    const uint64_t idx64 = (uint64_t)index;
    return mzk_merkle_open_batch(t, &idx64, 1, path, stride, path_len, depth);
  }
  MZK_ENTER();
  MerkleScope ms(t);
  MZK_TRY(ms.rc);
  hipStream_t s = ms.s;
  WsGuard wsg(s);          // WS_MISC_C below is shared with every other entry point of the context
  if (t->ragged) {
    // subtree p (depth D) that holds the leaf.  Merkle::open only terminates when its descent ends in a TWO-leaf
    // slice (merkle.rs:32-34); a one-leaf slice recurses forever (mid = 0), so those indices are an error here.
    size_t p = (size_t)(std::upper_bound(t->node_start.begin(), t->node_start.end(), (uint32_t)index) - t->node_start.begin()) - 1;
    const uint32_t first = t->node_start[p], cnt = t->node_start[p + 1] - first;
    const size_t sibp = p ^ 1;
    const uint32_t sib_cnt = t->node_start[sibp + 1] - t->node_start[sibp];
    if (cnt == 1 && sib_cnt != 1) {
      set_error("merkle_open: leaf %zu is the one-leaf half of a three-leaf slice; Merkle::open recurses forever there (merkle.rs:36-45)", index);
      return MZK_E_LENGTH;
    }
    // cnt == 2: the recursion ends inside subtree p (entry 0 = the other leaf), then climbs the item tree;
    // cnt == 1 with a one-leaf sibling: it ends one level higher, on the two-leaf slice {p, p ^ 1} -- the item path itself.
    int k = 0;
    if (cnt == 2) {
      const size_t other = first + (1 - (index - first));
      const size_t l0 = (size_t)(t->offsets[other + 1] - t->offsets[other]);
      if (stride < l0) { set_error("merkle_open: stride %zu < entry length %zu", stride, l0); return MZK_E_LENGTH; }
      if (l0) MZK_HIP(hipMemcpyAsync(path, (const uint8_t*)t->d_leaves + t->offsets[other], l0, hipMemcpyDeviceToHost, s));
      path_len[0] = l0;
      k = 1;
    }
    const size_t l1 = (size_t)(t->item_off[sibp + 1] - t->item_off[sibp]);
    if (stride < l1) { set_error("merkle_open: stride %zu < entry length %zu", stride, l1); return MZK_E_LENGTH; }
    if (l1) MZK_HIP(hipMemcpyAsync(path + (size_t)k * stride, (const uint8_t*)t->d_items + t->item_off[sibp], l1, hipMemcpyDeviceToHost, s));   // sibling item: a leaf or a digest
    path_len[k] = l1;
    if (t->depth > 1) {
      u64* d_out;
      MZK_TRY(ws_get(WS_MISC_C, (size_t)t->depth * 32, (void**)&d_out));
      hipLaunchKernelGGL(k_merkle_gather, dim3(1), dim3(64), 0, s, (const u64*)t->d_nodes, t->m, p, t->depth, d_out);
      MZK_HIP(hipGetLastError());
      uint8_t tmp[64 * 32];
      MZK_TRY(d2h_sync(tmp, d_out, (size_t)(t->depth - 1) * 32, s));
      for (int l = 1; l < t->depth; l++) { memcpy(path + (size_t)(l + k) * stride, tmp + 32 * (l - 1), 32); path_len[l + k] = 32; }
    } else {
      MZK_HIP(hipStreamSynchronize(s));
    }
    *depth = (size_t)t->depth + k;
    return MZK_OK;
  }
  const size_t sib = index ^ 1;
  // byte leaves: entry 0 is the sibling LEAF, verbatim
  {
    const size_t len = (size_t)(t->offsets[sib + 1] - t->offsets[sib]);
    if (stride < len) { set_error("merkle_open: stride %zu < leaf length %zu", stride, len); return MZK_E_LENGTH; }
    if (len) MZK_HIP(hipMemcpyAsync(path, (const uint8_t*)t->d_leaves + t->offsets[sib], len, hipMemcpyDeviceToHost, s));
    path_len[0] = len;
  }
  if (t->depth > 1) {
    u64* d_out;
    MZK_TRY(ws_get(WS_MISC_C, (size_t)t->depth * 32, (void**)&d_out));
    hipLaunchKernelGGL(k_merkle_gather, dim3(1), dim3(64), 0, s, (const u64*)t->d_nodes, t->n, index, t->depth, d_out);
    MZK_HIP(hipGetLastError());
    uint8_t tmp[64 * 32];
    MZK_TRY(d2h_sync(tmp, d_out, (size_t)(t->depth - 1) * 32, s));
    for (int l = 1; l < t->depth; l++) { memcpy(path + (size_t)l * stride, tmp + 32 * (l - 1), 32); path_len[l] = 32; }
  } else {
    MZK_HIP(hipStreamSynchronize(s));
  }
  *depth = (size_t)t->depth;
  return MZK_OK;
}

// `count` openings of one tree: paths[(q * depth + l) * stride ..] / path_lens[q * depth + l] as mzk_merkle_open writes them
// for index indices[q]; *depth entries per path.  Field-element trees with a power-of-two leaf count (every codeword of
// the provers): ONE gather launch and ONE copy for all paths.  Byte-leaf and ragged trees: MZK_E_ARG (open one by one).
int mzk_merkle_open_batch(const mzk_merkle* t, const uint64_t* indices, size_t count, uint8_t* paths, size_t stride, uint64_t* path_lens,
                          size_t* depth) {
  if (!t || !depth || ((!indices || !paths || !path_lens) && count)) { set_error("merkle_open_batch: null pointer"); return MZK_E_ARG; }
  if (t->n < 2) { set_error("merkle_open: needs at least two leaves (merkle.rs:32)"); return MZK_E_LENGTH; }
  if (t->ragged || t->kind != 0) {
    set_error("merkle_open_batch: field-element trees with a power-of-two leaf count only (this one: %s); open its paths one by one",
              t->kind != 0 ? "byte leaves" : "ragged");
    return MZK_E_ARG;
  }
  *depth = (size_t)t->depth;
  if (count == 0) return MZK_OK;
  if (stride < 32) { set_error("merkle_open: stride < 32"); return MZK_E_LENGTH; }
  for (size_t q = 0; q < count; q++)
    if (indices[q] >= t->n) { set_error("merkle_open: index %llu out of range", (unsigned long long)indices[q]); return MZK_E_LENGTH; }
  MZK_ENTER();
  MerkleScope ms(t);
  MZK_TRY(ms.rc);
  hipStream_t s = ms.s;
  WsGuard wsg(s);
  const int lw = (int)field_words(t->field), nl = field_limbs64(t->field);
  const size_t node_words64 = count * (size_t)(t->depth - 1) * 4;
  u64 *d_idx, *d_on;
  u32* d_ol;
  MZK_TRY(ws_get(WS_MISC_C, count * 8 + node_words64 * 8 + count * (size_t)lw * 4 + 64, (void**)&d_idx));
  d_on = d_idx + count;
  d_ol = (u32*)(d_on + node_words64);
  MZK_HIP(hipMemcpyAsync(d_idx, indices, count * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_merkle_gather_batch, dim3((unsigned)count), dim3(64), 0, s, (const u64*)t->d_nodes, (const u32*)t->d_leaves, lw, t->n,
                     (const u64*)d_idx, t->depth, d_on, d_ol);
  MZK_HIP(hipGetLastError());
  std::vector<uint64_t> hn(node_words64 + 1);
  std::vector<uint32_t> hl(count * (size_t)lw);
  if (node_words64) MZK_HIP(hipMemcpyAsync(hn.data(), d_on, node_words64 * 8, hipMemcpyDeviceToHost, s));
  MZK_TRY(d2h_sync(hl.data(), d_ol, count * (size_t)lw * 4, s));
  for (size_t q = 0; q < count; q++) {
    uint64_t limbs[4] = {0, 0, 0, 0};
    memcpy(limbs, hl.data() + q * lw, (size_t)lw * 4);
    uint8_t buf[48];
    const size_t sib = (size_t)indices[q] ^ 1;
    const size_t len = host_bincode_field(limbs, nl, buf, !t->neg.empty() && t->neg[sib]);
    if (stride < len) { set_error("merkle_open: stride %zu < leaf length %zu", stride, len); return MZK_E_LENGTH; }
    uint8_t* pq = paths + q * (size_t)t->depth * stride;
    memcpy(pq, buf, len);
    path_lens[q * (size_t)t->depth] = len;
    for (int l = 1; l < t->depth; l++) {
      memcpy(pq + (size_t)l * stride, hn.data() + (q * (size_t)(t->depth - 1) + (l - 1)) * 4, 32);
      path_lens[q * (size_t)t->depth + l] = 32;
    }
  }
  return MZK_OK;
}

// Openings of SEVERAL trees in one call: the query phase of FRI::prove (fri.rs:127-137, reveal :211-260) opens the a / b indices
// in round i's tree and the c indices in round i+1's, for every round -- with the trees of mzk_fri_commit_keep_trees that is one
// call, one copy back and one synchronisation instead of two of each per round (48 us per call measured).  Tree t takes the
// next counts[t] entries of `indices`; its paths follow tree t-1's in paths / path_lens, laid out as mzk_merkle_open_batch
// lays them out with depths[t] entries per path.  counts[t] == 0 skips tree t (which may then be NULL).
int mzk_merkle_open_multi(const mzk_merkle* const* trees, size_t n_trees, const uint64_t* indices, const size_t* counts, uint8_t* paths,
                          size_t stride, uint64_t* path_lens, size_t* depths) {
  if (!trees || !counts || !depths) { set_error("merkle_open_multi: null pointer"); return MZK_E_ARG; }
  size_t total = 0, total_nodes = 0, total_leaf_words = 0;
  const mzk_merkle* first = nullptr;
  for (size_t t = 0; t < n_trees; t++) {
    depths[t] = trees[t] ? (size_t)trees[t]->depth : 0;
    if (counts[t] == 0) continue;
    const mzk_merkle* tr = trees[t];
    if (!tr || !indices || !paths || !path_lens) { set_error("merkle_open_multi: null pointer"); return MZK_E_ARG; }
    if (tr->n < 2) { set_error("merkle_open: needs at least two leaves (merkle.rs:32)"); return MZK_E_LENGTH; }
    if (tr->ragged || tr->kind != 0) {
      set_error("merkle_open_multi: field-element trees with a power-of-two leaf count only (tree %zu: %s)", t, tr->kind != 0 ? "byte leaves" : "ragged");
      return MZK_E_ARG;
    }
    for (size_t q = 0; q < counts[t]; q++)
      if (indices[total + q] >= tr->n) { set_error("merkle_open: index %llu out of range (tree %zu)", (unsigned long long)indices[total + q], t); return MZK_E_LENGTH; }
    if (!first) first = tr;
    if (tr->ctx_index != first->ctx_index || tr->device != first->device) {
      set_error("merkle_open_multi: tree %zu lives on context %d, the first opened tree on context %d; one call opens trees of one context", t,
                tr->ctx_index, first->ctx_index);
      return MZK_E_ARG;
    }
    total += counts[t];
    total_nodes += counts[t] * (size_t)(tr->depth - 1) * 4;
    total_leaf_words += counts[t] * field_words(tr->field);
  }
  if (total == 0) return MZK_OK;
  if (stride < 32) { set_error("merkle_open: stride < 32"); return MZK_E_LENGTH; }
  MZK_ENTER();
  MerkleScope ms(first);
  MZK_TRY(ms.rc);
  hipStream_t s = ms.s;
  for (size_t t = 0; t < n_trees; t++)      // every opened tree's build, not only the first one's
    if (counts[t] && trees[t]->built) MZK_HIP(hipStreamWaitEvent(s, trees[t]->built, 0));
  WsGuard wsg(s);
  u64 *d_idx, *d_on;
  u32* d_ol;
  MZK_TRY(ws_get(WS_MISC_C, total * 8 + total_nodes * 8 + total_leaf_words * 4 + 64, (void**)&d_idx));
  d_on = d_idx + total;
  d_ol = (u32*)(d_on + total_nodes);
  MZK_HIP(hipMemcpyAsync(d_idx, indices, total * 8, hipMemcpyHostToDevice, s));
  {
    size_t at = 0, at_nodes = 0, at_leaf = 0;
    for (size_t t = 0; t < n_trees; t++) {
      if (counts[t] == 0) continue;
      const mzk_merkle* tr = trees[t];
      const int lw = (int)field_words(tr->field);
      hipLaunchKernelGGL(k_merkle_gather_batch, dim3((unsigned)counts[t]), dim3(64), 0, s, (const u64*)tr->d_nodes, (const u32*)tr->d_leaves, lw, tr->n,
                         (const u64*)(d_idx + at), tr->depth, d_on + at_nodes, d_ol + at_leaf);
      at += counts[t];
      at_nodes += counts[t] * (size_t)(tr->depth - 1) * 4;
      at_leaf += counts[t] * (size_t)lw;
    }
  }
  MZK_HIP(hipGetLastError());
  std::vector<uint64_t> hn(total_nodes + 1);
  std::vector<uint32_t> hl(total_leaf_words + 1);
  if (total_nodes) MZK_HIP(hipMemcpyAsync(hn.data(), d_on, total_nodes * 8, hipMemcpyDeviceToHost, s));
  MZK_TRY(d2h_sync(hl.data(), d_ol, total_leaf_words * 4, s));
  size_t at = 0, at_nodes = 0, at_leaf = 0, at_entry = 0;
  for (size_t t = 0; t < n_trees; t++) {
    if (counts[t] == 0) continue;
    const mzk_merkle* tr = trees[t];
    const int lw = (int)field_words(tr->field), nl = field_limbs64(tr->field);
    for (size_t q = 0; q < counts[t]; q++) {
      uint64_t limbs[4] = {0, 0, 0, 0};
      memcpy(limbs, hl.data() + at_leaf + q * lw, (size_t)lw * 4);
      uint8_t buf[48];
      const size_t sib = (size_t)indices[at + q] ^ 1;
      const size_t len = host_bincode_field(limbs, nl, buf, !tr->neg.empty() && tr->neg[sib]);
      if (stride < len) { set_error("merkle_open: stride %zu < leaf length %zu", stride, len); return MZK_E_LENGTH; }
      uint8_t* pq = paths + (at_entry + q * (size_t)tr->depth) * stride;
      memcpy(pq, buf, len);
      path_lens[at_entry + q * (size_t)tr->depth] = len;
      for (int l = 1; l < tr->depth; l++) {
        memcpy(pq + (size_t)l * stride, hn.data() + at_nodes + (q * (size_t)(tr->depth - 1) + (l - 1)) * 4, 32);
        path_lens[at_entry + q * (size_t)tr->depth + l] = 32;
      }
    }
    at += counts[t];
    at_nodes += counts[t] * (size_t)(tr->depth - 1) * 4;
    at_leaf += counts[t] * (size_t)lw;
    at_entry += counts[t] * (size_t)tr->depth;
  }
  return MZK_OK;
}

void mzk_merkle_free(mzk_merkle* t) {
  if (!t) return;
  if (t->shared) {
    if (--t->shared->refs == 0) { (void)hipFree(t->shared->p); delete t->shared; }
    if (!t->owns_leaves) t->d_leaves = nullptr;
    t->d_nodes = nullptr;
  }
  if (t->d_leaves) (void)hipFree(t->d_leaves);
  if (t->d_nodes) (void)hipFree(t->d_nodes);
  if (t->d_items) (void)hipFree(t->d_items);
  if (t->built) (void)hipEventDestroy(t->built);
  delete t;
}

// One-shot commit of a device-resident codeword (the FRI round: root -> transcript -> alpha -> fold); nothing is
// retained, the node levels live in workspace.  root: 32 bytes (n >= 2) or the leaf bytes (n == 1; cap >= 41).
int mzk_merkle_commit_field_dev(int field_id, const void* d_elems, size_t n, uint8_t* root, size_t cap, size_t* root_len, void* stream) {
  MZK_ENTER();
  WsGuard wsg((hipStream_t)stream);
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("merkle: bad field id %d", field_id); return MZK_E_ARG; }
  if (n == 0) { set_error("merkle: empty leaf set (Merkle::commit recurses forever on it, merkle.rs:20-22)"); return MZK_E_LENGTH; }
  if (!d_elems || !root || !root_len) { set_error("merkle: null pointer"); return MZK_E_ARG; }
  if (!is_pow2(n)) {      // ragged (merkle.rs:15-25 accepts it; no prover call site produces it): handle path
    mzk_merkle* t = nullptr;
    MZK_TRY(merkle_build(0, field_id, d_elems, true, n * field_bytes(field_id), nullptr, n, &t, (hipStream_t)stream));
    const int rc = mzk_merkle_root(t, root, cap, root_len);
    mzk_merkle_free(t);
    return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  if (n == 1) {
    uint64_t limbs[4];
    uint8_t buf[48];
    MZK_TRY(d2h_sync(limbs, d_elems, field_bytes(field_id), s));
    const size_t len = host_bincode_field(limbs, field_limbs64(field_id), buf);
    if (cap < len) { set_error("merkle: root buffer too small"); return MZK_E_LENGTH; }
    memcpy(root, buf, len);
    *root_len = len;
    return MZK_OK;
  }
  if (cap < 32) { set_error("merkle: root buffer too small"); return MZK_E_LENGTH; }
  u64* d_nodes;
  MZK_TRY(ws_get(WS_MERKLE_NODES, (n - 1) * 32, (void**)&d_nodes));
  RootMailbox mb;
  bool mailed = false;
  MZK_TRY(merkle_mailbox(&mb));
  MZK_TRY(merkle_hash_levels(0, field_id, d_elems, nullptr, n, d_nodes, s, nullptr, 1, mb.p, mb.seq, &mailed));
  MZK_TRY(merkle_root_to_host(root, d_nodes + 4 * (n - 2), mb, mailed, s));
  *root_len = 32;
  return MZK_OK;
}
// Merkle::commit of `batch` codewords of n elements each (n a power of two >= 2), roots only: one set of launches for all
// trees, so the latency-bound upper levels of the trees run side by side.
int mzk_merkle_commit_field_batch_dev(int field_id, const void* d_elems, size_t n, size_t batch, uint8_t* roots, void* stream) {
  MZK_ENTER();
  WsGuard wsg((hipStream_t)stream);
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("merkle: bad field id %d", field_id); return MZK_E_ARG; }
  if (batch == 0) return MZK_OK;
  if (n == 0) { set_error("merkle: empty leaf set (Merkle::commit recurses forever on it, merkle.rs:20-22)"); return MZK_E_LENGTH; }
  if (!d_elems || !roots) { set_error("merkle: null pointer"); return MZK_E_ARG; }
  if (!is_pow2(n) || n < 2) { set_error("merkle batch: codewords of %zu elements (need a power of two >= 2; use the single-tree calls)", n); return MZK_E_NOT_POW2; }
  if (batch > ((size_t)1 << 36) / n) { set_error("merkle batch: too many leaves"); return MZK_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const size_t total = n * batch;
  u64* d_nodes;
  MZK_TRY(ws_get(WS_MERKLE_NODES, (total - batch) * 32, (void**)&d_nodes));
  MZK_TRY(merkle_hash_levels(0, field_id, d_elems, nullptr, total, d_nodes, s, nullptr, batch));
  MZK_TRY(d2h_sync(roots, d_nodes + 4 * (total - 2 * batch), batch * 32, s));
  return MZK_OK;
}
int mzk_merkle_commit_field_batch(int field_id, const uint64_t* elems, size_t n, size_t batch, uint8_t* roots) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("merkle: bad field id %d", field_id); return MZK_E_ARG; }
  if (batch == 0) return MZK_OK;
  if (!elems) { set_error("merkle: null pointer"); return MZK_E_ARG; }
  hipStream_t s = ctx().stream;
  void* d;
  {
    WsGuard wsg(s);
    MZK_TRY(ws_get(WS_MISC_D, n * batch * field_bytes(field_id) + 16, &d));
    MZK_HIP(hipMemcpyAsync(d, elems, n * batch * field_bytes(field_id), hipMemcpyHostToDevice, s));
  }
  return mzk_merkle_commit_field_batch_dev(field_id, d, n, batch, roots, s);
}
int mzk_merkle_commit_field(int field_id, const uint64_t* elems, size_t n, uint8_t* root, size_t cap, size_t* root_len) {
  mzk_merkle* t = nullptr;
  MZK_TRY(mzk_merkle_build_field(field_id, elems, n, &t));
  const int rc = mzk_merkle_root(t, root, cap, root_len);
  mzk_merkle_free(t);
  return rc;
}
int mzk_merkle_commit_bytes(const uint8_t* leaves, const uint64_t* offsets, size_t n, uint8_t* root, size_t cap, size_t* root_len) {
  mzk_merkle* t = nullptr;
  MZK_TRY(mzk_merkle_build_bytes(leaves, offsets, n, &t));
  const int rc = mzk_merkle_root(t, root, cap, root_len);
  mzk_merkle_free(t);
  return rc;
}

// FRI::commit (zkstark/fri.rs:144-209) with the codewords resident in HBM for the whole loop.  Per round r:
//   root_r = Merkle::commit(codeword_r as bincode leaves)              fri.rs:160-168
//   challenge(user, r, last, root_r, len, alpha)  -- the host pushes the root to its proof stream and, unless
//                                                    `last`, samples alpha from it                 fri.rs:168-176
//   codeword_{r+1} = split-and-fold(codeword_r, alpha, offset, omega); omega, offset squared      fri.rs:182-195
// Outputs: roots (num_rounds entries of 48 bytes, lengths in root_len: 32, or the leaf bytes once a codeword has
// shrunk to one element) and all num_rounds codewords concatenated (n + n/2 + ... elements) -- the reference's
// return value `(codewords, roots)`; sending the last codeword (fri.rs:198-206) is the caller's transcript work.
static int fri_commit_rounds(int field_id, const uint64_t* codeword, const uint8_t* negative, size_t n, const uint64_t* omega, const uint64_t* offset,
                             int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out,
                             mzk_merkle** trees_out, bool on_device);
// trees_out (optional, num_rounds entries): the Merkle tree of every round's codeword stays on the device as a handle for the
// query phase (mzk_merkle_open_batch; fri.rs:211-260 opens from exactly these codewords); a one-element round gets NULL.
// On failure every handle made so far is released.
static int fri_commit_impl(int field_id, const uint64_t* codeword, const uint8_t* negative, size_t n, const uint64_t* omega, const uint64_t* offset,
                           int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out,
                           mzk_merkle** trees_out = nullptr, bool on_device = false) {
  if (trees_out) for (int r = 0; r < num_rounds; r++) trees_out[r] = nullptr;
  const int rc = fri_commit_rounds(field_id, codeword, negative, n, omega, offset, num_rounds, challenge, user, roots, root_len, codewords_out, trees_out, on_device);
  if (rc != MZK_OK && trees_out)
    for (int r = 0; r < num_rounds; r++) { if (trees_out[r]) mzk_merkle_free(trees_out[r]); trees_out[r] = nullptr; }
  return rc;
}
static int fri_commit_rounds(int field_id, const uint64_t* codeword, const uint8_t* negative, size_t n, const uint64_t* omega, const uint64_t* offset,
                             int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out,
                             mzk_merkle** trees_out, bool on_device) {
  MZK_ENTER();
  if (field_id != MZK_FIELD_FR && field_id != MZK_FIELD_M128) { set_error("fri_commit: bad field id %d", field_id); return MZK_E_ARG; }
  if (num_rounds <= 0) return MZK_OK;
  if (!codeword || !omega || !offset || !challenge || !roots || !root_len) { set_error("fri_commit: null pointer"); return MZK_E_ARG; }
  if (!codewords_out && !trees_out) { set_error("fri_commit: no codeword output and no trees kept"); return MZK_E_ARG; }
  if (n == 0) { set_error("fri_commit: empty codeword"); return MZK_E_LENGTH; }
  if (!is_pow2(n)) { set_error("fri_commit: codeword length must be a power of two"); return MZK_E_NOT_POW2; }
  if ((n >> (num_rounds - 1)) == 0) { set_error("fri_commit: %d rounds halve a length-%zu codeword away", num_rounds, n); return MZK_E_LENGTH; }
  const HostField* hf = host_field(field_id);
  if (!h_is_canonical(hf, omega) || !h_is_canonical(hf, offset)) { set_error("fri_commit: parameter not canonical"); return MZK_E_RANGE; }
  hipStream_t s = ctx().stream;
  WsGuard wsg(s);
  const size_t esz = field_bytes(field_id);
  const int nl = hf->nl;
  size_t total = 0;
  for (int r = 0; r < num_rounds; r++) total += n >> r;
  uint8_t* d_all;
  u64* d_nodes;
  mzk_merkle::SharedBlock* shared = nullptr;
  if (trees_out) {
    // trees kept: codewords and digests of every round in ONE allocation that the rounds' handles share (2 hipMalloc and two
    // device copies per round before: ~25 us of a ~155-us round at 2^14); layout: [codewords of all rounds | digests of round 0 | 1 | ...]
    for (int r = 0; r < num_rounds; r++) trees_out[r] = nullptr;
    size_t node_bytes = 0;
    for (int r = 0; r < num_rounds; r++) if ((n >> r) >= 2) node_bytes += ((n >> r) - 1) * 32;
    const size_t cw_bytes = (total * esz + 255) & ~(size_t)255;
    void* blk = nullptr;
    MZK_TRY(dev_alloc(&blk, cw_bytes + node_bytes + 256, "fri_commit: kept trees"));
    shared = new mzk_merkle::SharedBlock{blk, 1};        // this function's own reference, dropped at the end
    d_all = (uint8_t*)blk;
    d_nodes = (u64*)((uint8_t*)blk + cw_bytes);
  } else {
    MZK_TRY(ws_get(WS_NTT_IO_A, total * esz, (void**)&d_all));
    MZK_TRY(ws_get(WS_MERKLE_NODES, n * 32, (void**)&d_nodes));
  }
  struct DropRef {        // every exit path: the handles already handed out keep the block alive, otherwise it goes
    mzk_merkle::SharedBlock* b;
    ~DropRef() { if (b && --b->refs == 0) { (void)hipFree(b->p); delete b; } }
  } drop{shared};
  MZK_HIP(hipMemcpyAsync(d_all, codeword, n * esz, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  u8* d_neg = nullptr;
  if (negative) {     // round 0 hashes bincode of the UNSANITIZED elements (fri.rs:160-166); the fold sanitizes (fri.rs:190)
    MZK_TRY(ws_get(WS_MISC_E, n, (void**)&d_neg));
    MZK_HIP(hipMemcpyAsync(d_neg, negative, n, hipMemcpyHostToDevice, s));
  }
  uint64_t om[4], of[4], alpha[4];
  memcpy(om, omega, 8 * nl);
  memcpy(of, offset, 8 * nl);
  // the fold's constants 2^-1, offset^-1, omega^-1: inverted ONCE here and squared along with omega and offset (three host
  // inversions per round before)
  FriFoldConsts fc;
  MZK_TRY(fri_fold_consts(field_id, of, om, &fc));
  uint8_t* cur = d_all;
  size_t len = n;
  for (int r = 0; r < num_rounds; r++) {
    uint8_t* root = roots + 48 * (size_t)r;
    if (len == 1) {
      uint64_t limbs[4];
      MZK_TRY(d2h_sync(limbs, cur, esz, s));
      root_len[r] = host_bincode_field(limbs, nl, root, r == 0 && negative && negative[0]);
    } else {
      RootMailbox mb;
      bool mailed = false;
      MZK_TRY(merkle_mailbox(&mb));
      MZK_TRY(merkle_hash_levels(0, field_id, cur, nullptr, len, d_nodes, s, r == 0 ? d_neg : nullptr, 1, mb.p, mb.seq, &mailed));
      if (trees_out) {      // keep this round's tree: its leaves and digests stay where they were computed, in the shared block
        mzk_merkle* t = new mzk_merkle();
        t->kind = 0; t->field = field_id; t->n = len; t->stream = s;
        t->d_nodes = d_nodes; t->d_leaves = cur;
        t->shared = shared; shared->refs++;
        t->depth = 0;
        while (((size_t)1 << t->depth) < len) t->depth++;
        trees_out[r] = t;
        if (r == 0 && negative) {
          // round 0 commits to the UNSANITIZED elements and the fold below canonicalises the codeword in place: the tree keeps
          // the magnitudes as they were given, in a copy of its own
          void* own = nullptr;
          MZK_TRY(dev_alloc(&own, len * esz, "fri_commit: round-0 leaves"));
          MZK_HIP(hipMemcpyAsync(own, cur, len * esz, hipMemcpyDeviceToDevice, s));
          t->d_leaves = own; t->owns_leaves = true;
          t->neg.assign(negative, negative + n);
        }
        MZK_TRY(merkle_stamp(t, s, true));
      }
      MZK_TRY(merkle_root_to_host(root, d_nodes + 4 * (len - 2), mb, mailed, s));      // the transcript needs the root now
      root_len[r] = 32;
    }
    if (r == 0 && d_neg) {      // from here on the codeword is its canonical representative: v -> p - |v| where Sign::Minus
      if (field_id == MZK_FIELD_M128) hipLaunchKernelGGL((k_canonicalize_signed<4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (u32*)cur, (const u8*)d_neg, n, field_id);
      else hipLaunchKernelGGL((k_canonicalize_signed<8>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (u32*)cur, (const u8*)d_neg, n, field_id);
      MZK_HIP(hipGetLastError());
    }
    const int last = (r == num_rounds - 1);
    memset(alpha, 0xff, sizeof alpha);     // a callback that forgets alpha leaves a non-canonical value, never a silent 0
    const int crc = challenge(user, r, last, root, (size_t)root_len[r], alpha);
    if (crc != 0) { set_error("fri_commit: challenge callback failed in round %d (status %d)", r, crc); return MZK_E_CALLBACK; }
    if (last) break;
    if (!h_is_canonical(hf, alpha)) { set_error("fri_commit: challenge of round %d not canonical", r); return MZK_E_RANGE; }
    uint8_t* next = cur + len * esz;
    MZK_TRY(fri_fold_dev_consts(field_id, cur, len, alpha, fc, next, s));
    h_mulmod(hf, om, om, om);
    h_mulmod(hf, of, of, of);
    fri_fold_consts_square(field_id, &fc);
    if (trees_out && len >= 2) d_nodes += 4 * (len - 1);        // the next round's digests follow this round's
    cur = next;
    len /= 2;
  }
  MZK_TRY(d2h_sync(codewords_out, d_all, codewords_out ? total * esz : 0, s));
  return MZK_OK;
}
int mzk_fri_commit(int field_id, const uint64_t* codeword, size_t n, const uint64_t* omega, const uint64_t* offset, int num_rounds,
                   mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out) {
  return fri_commit_impl(field_id, codeword, nullptr, n, omega, offset, num_rounds, challenge, user, roots, root_len, codewords_out);
}
int mzk_fri_commit_signed(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, const uint64_t* omega, const uint64_t* offset,
                          int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out) {
  return fri_commit_impl(field_id, magnitudes, negative, n, omega, offset, num_rounds, challenge, user, roots, root_len, codewords_out);
}
int mzk_fri_commit_keep_trees(int field_id, const uint64_t* magnitudes, const uint8_t* negative, size_t n, const uint64_t* omega, const uint64_t* offset,
                              int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out,
                              mzk_merkle** trees_out) {
  if (!trees_out && num_rounds > 0) { set_error("fri_commit_keep_trees: null handle array"); return MZK_E_ARG; }
  return fri_commit_impl(field_id, magnitudes, negative, n, omega, offset, num_rounds, challenge, user, roots, root_len, codewords_out, trees_out);
}
// the same with the initial codeword already in HBM (the output of mzk_coset_lde_dev: fast_stark.rs commits to what it has just
// extended); d_codeword must be complete before the call.  negative (host, optional) as in mzk_fri_commit_signed.
int mzk_fri_commit_keep_trees_dev(int field_id, const void* d_magnitudes, const uint8_t* negative, size_t n, const uint64_t* omega, const uint64_t* offset,
                                  int num_rounds, mzk_fri_challenge_fn challenge, void* user, uint8_t* roots, uint64_t* root_len, uint64_t* codewords_out,
                                  mzk_merkle** trees_out) {
  if (!trees_out && num_rounds > 0) { set_error("fri_commit_keep_trees: null handle array"); return MZK_E_ARG; }
  return fri_commit_impl(field_id, (const uint64_t*)d_magnitudes, negative, n, omega, offset, num_rounds, challenge, user, roots, root_len, codewords_out, trees_out, true);
}
// The elements a tree was built over, at `count` positions: magnitudes (count x limbs) and, if wanted, their Sign::Minus flags --
// what FRI::reveal sends next to the authentication paths (fri.rs:224-233), without the codewords ever leaving the device.
int mzk_merkle_leaves(const mzk_merkle* t, const uint64_t* indices, size_t count, uint64_t* magnitudes, uint8_t* negative) {
  if (!t || ((!indices || !magnitudes) && count)) { set_error("merkle_leaves: null pointer"); return MZK_E_ARG; }
  if (t->kind != 0) { set_error("merkle_leaves: field-element trees only"); return MZK_E_ARG; }
  for (size_t q = 0; q < count; q++)
    if (indices[q] >= t->n) { set_error("merkle_leaves: index %llu out of range", (unsigned long long)indices[q]); return MZK_E_LENGTH; }
  if (count == 0) return MZK_OK;
  MZK_ENTER();
  MerkleScope ms(t);
  MZK_TRY(ms.rc);
  hipStream_t s = ms.s;
  WsGuard wsg(s);
  const size_t esz = field_bytes(t->field);
  const int lw = (int)field_words(t->field);
  u64* d_idx;
  MZK_TRY(ws_get(WS_MISC_C, count * 8 + count * esz + 64, (void**)&d_idx));
  u32* d_out = (u32*)(d_idx + count);
  MZK_HIP(hipMemcpyAsync(d_idx, indices, count * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_gather_leaves, dim3((unsigned)((count * lw + 255) / 256)), dim3(256), 0, s, (const u32*)t->d_leaves, lw, (const u64*)d_idx, count, d_out);
  MZK_HIP(hipGetLastError());
  MZK_HIP(hipMemcpyAsync(magnitudes, d_out, count * esz, hipMemcpyDeviceToHost, s));
  if (negative) for (size_t q = 0; q < count; q++) negative[q] = (!t->neg.empty() && t->neg[(size_t)indices[q]]) ? 1 : 0;
  MZK_HIP(hipStreamSynchronize(s));
  return MZK_OK;
}

}  // extern "C"
