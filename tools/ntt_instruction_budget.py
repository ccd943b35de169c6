#!/usr/bin/env python3
"""Instruction budget of one pass of the 2^20 transforms (2^10 levels per 4096-element tile, 1024 lanes, four elements per lane),
by category, from the ISA:

  1. tools/microbench/ntt_op_costs.hip holds one kernel per elementary operation (product, butterfly sums, carry pass, pack /
     unpack, reductions); it is compiled for gfx950 and disassembled here, and each operation's VALU instructions are its
     kernel's minus the `base` kernel's (same loads and stores, no arithmetic).  Half-rate instructions (v_mad_*64*, 64-bit
     shifts / adds, v_mul_lo / hi) are counted separately: they take two issue slots (profiles/r01_ubench_instr_rates.txt).
  2. The dynamic operation counts per lane and pass follow from the stage schedule (five radix-4 stage pairs; groups whose twiddles
     are all 1 skip three of four products; the strided pass multiplies by the inter-pass twiddle).
  3. What is left of the hardware counter (SQ_INSTS_VALU per wave, from profiles/) is addressing, loop control and twiddle handling.

    python tools/ntt_instruction_budget.py [m128_strided=<SQ_INSTS_VALU per launch>] [m128_last=...] [fr_strided=...] [fr_last=...]
"""
import os, re, subprocess, sys, tempfile, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "microbench", "ntt_op_costs.hip")
HALF = re.compile(r"^(v_mad_[iu]64_[iu]32|v_lshrrev_b64|v_ashrrev_i64|v_lshlrev_b64|v_lshl_add_u64|v_mul_lo_u32|v_mul_hi_[iu]32|v_addc_co_u32|v_subb_co_u32|v_mad_u32_u24|v_mul_u32_u24|v_mul_i32_i24|v_mad_i32_i24)")


def probe_costs():
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "ops.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", s, SRC],
                              stderr=subprocess.DEVNULL)
        txt = open(s).read()
    out = {}
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)s_endpgm", txt, re.S | re.M):
        full = half = 0
        for line in m.group(2).splitlines():
            op = line.strip().split(" ")[0] if line.strip() else ""
            if op.startswith("v_"):
                if HALF.match(op):
                    half += 1
                else:
                    full += 1
        out[m.group(1)] = (full, half)
    return out


def main():
    c = probe_costs()
    def cost(name, base):
        return (c[name][0] - c[base][0], c[name][1] - c[base][1])
    ops = collections.OrderedDict()
    for n in ("m128_smul", "m128_mul", "m128_sbfly", "m128_scarry", "m128_sbias", "m128_sreduce", "m128_bfly_lazy_r4", "m128_bfly_carry_r4", "m128_reduce_r4", "m128_weak_reduce_r4"):
        ops[n] = cost(n, "m128_base")
    ops["m128_unpack"] = cost("m128_unpack", "m128_unpack_base")
    ops["m128_pack"] = cost("m128_pack", "m128_pack_base")
    for n in ("fr_mul", "fr_shoup", "fr_bfly_lazy", "fr_bfly_carry", "fr_reduce", "fr_weak_reduce", "fr_fit"):
        ops[n] = cost(n, "fr_base")
    ops["fr_unpack"] = cost("fr_unpack", "fr_unpack_base")
    ops["fr_pack"] = cost("fr_pack", "fr_pack_base")
    print("== VALU instructions per operation (full rate + half rate), from the ISA of tools/microbench/ntt_op_costs.hip")
    for n, (f, h) in ops.items():
        print("  %-22s %4d = %3d + %3d half-rate   (%d issue slots)" % (n, f + h, f, h, f + 2 * h))
    sq = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    tot = lambda n, k=1: tuple(k * v for v in ops[n])
    # dynamic counts per LANE and pass: four elements, five stage pairs = 20 butterflies; products: pair 1 one (zeta), pairs at
    # s = 3, 5 skip three of four products in 1/4 and 1/16 of the waves, pairs s = 7, 9 multiply everywhere (a wave mixes j1 there)
    prods = 1 + (0.25 * 1 + 0.75 * 4) + (1 / 16 * 1 + 15 / 16 * 4) + 4 + 4
    def show(title, rows, counter_key):
        print("\n== %s: per lane (4 elements) and pass" % title)
        s_f = s_h = 0
        for label, k, (f, h) in rows:
            print("  %-58s %6.1f x %3d+%3dh = %7.1f" % (label, k, f, h, k * (f + h)))
            s_f += k * f; s_h += k * h
        print("  %-58s %25.1f  (%.1f per element)" % ("accounted", s_f + s_h, (s_f + s_h) / 4))
        if counter_key in sq:
            per_wave = float(sq[counter_key]) / 4096
            print("  %-58s %25.1f  (%.1f per element)" % ("SQ_INSTS_VALU per wave (hardware counter)", per_wave, per_wave / 4))
            print("  %-58s %25.1f  (%.1f per element)" % ("rest: LDS / global addressing, twiddle indices, loop control", per_wave - s_f - s_h, (per_wave - s_f - s_h) / 4))
    show("M128 strided pass, this tree (signed lazy, sparse product)", [
        ("products, stage pairs (FeAsm::smul)", prods, ops["m128_smul"]),
        ("products, inter-pass twiddle", 4, ops["m128_smul"]),
        ("butterfly sums (fe_sadd + fe_ssub)", 20, ops["m128_sbfly"]),
        ("carry of the unmultiplied inputs (2 per group, pairs 2..5)", 8 + 2 * (0.25 + 1 / 16), ops["m128_scarry"]),
        ("bias before the inter-pass product", 4, ops["m128_sbias"]),
        ("unpack: data + inter-pass twiddles", 8, ops["m128_unpack"]),
        ("pack", 4, ops["m128_pack"])], "m128_strided")
    show("M128 last pass, this tree", [
        ("products, stage pairs (FeAsm::smul)", prods, ops["m128_smul"]),
        ("butterfly sums (fe_sadd + fe_ssub)", 20, ops["m128_sbfly"]),
        ("carry of the unmultiplied inputs", 8 + 2 * (0.25 + 1 / 16), ops["m128_scarry"]),
        ("canonical reduction (fe_sreduce, fast path)", 4, ops["m128_sreduce"]),
        ("unpack", 4, ops["m128_unpack"]),
        ("pack", 4, ops["m128_pack"])], "m128_last")
    old_mul = (19, 49)      # round 4's FeAsm<M128Params>::mul: 35 v_mad_u64_u32 + 5 v_mul_lo + 9 v_lshrrev_b64 half rate, 9 v_and + v_mov full
    show("M128 strided pass, ROUND 4 (for comparison: dense-modulus product, first stage of a pair limb-wise + 8p, second carrying)", [
        ("products, stage pairs", prods, old_mul),
        ("products, inter-pass twiddle", 4, old_mul),
        ("butterfly sums, first stage of a pair (fe_add + fe_sub<8>)", 10, ops["m128_bfly_lazy_r4"]),
        ("butterfly sums, second stage (fe_add_carry + fe_sub_carry<8>)", 10, ops["m128_bfly_carry_r4"]),
        ("weak reduction where a twiddle is 1", 3 * (0.25 + 1 / 16), ops["m128_weak_reduce_r4"]),
        ("unpack: data + inter-pass twiddles", 8, ops["m128_unpack"]),
        ("pack", 4, ops["m128_pack"])], "m128_strided_r04")
    show("BN254 Fr strided pass (Shoup products in stage pairs 1, 3, 5; Montgomery in 7, 9 and for the inter-pass twiddle)", [
        ("products, wave-uniform twiddles (shoup_mul, scalar constants)", 1 + 3.25 + 3.8125, ops["fr_shoup"]),
        ("products, LDS twiddles + inter-pass (FeAsm::mul)", 12, ops["fr_mul"]),
        ("butterfly sums, first stage of a pair", 10, ops["fr_bfly_lazy"]),
        ("butterfly sums, second stage (carrying)", 10, ops["fr_bfly_carry"]),
        ("weak reduction where a twiddle is 1", 3 * (0.25 + 1 / 16), ops["fr_weak_reduce"]),
        ("conditional subtraction before the store", 4, ops["fr_fit"]),
        ("unpack: data + inter-pass twiddles", 8, ops["fr_unpack"]),
        ("pack", 4, ops["fr_pack"])], "fr_strided")


if __name__ == "__main__":
    main()
