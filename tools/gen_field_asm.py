#!/usr/bin/env python3
"""Generate myzkp_amd/csrc/mzk_field_asm.h: the Montgomery product / square / two-product sum of mzk_field.h
(fe_mul, fe_sqr, fe_mul_add2 -- same column order, same 64-bit column sums, bit-identical results) as ONE inline-asm
block per call for the throughput kernels.

Why: hipcc reassociates every product-scanning column so that the carry of the previous column is added LAST (one
v_lshl_add_u64 per column), starts the columns from v_mov_b64 zeros and keeps a few copies; per 9-limb product that is
~255 instructions for 162 multiply-adds.  Here every column is one chain of v_mad_u64_u32 whose first addend is the
shifted previous column: 162 multiply-adds + 17 v_lshrrev_b64 + 26 v_and + 9 v_mul_lo = 214 instructions.  The chain is
serial (no instruction-level parallelism inside a lane), so it only pays where several waves per SIMD hide the
latency: the bucket accumulation of the MSM and the NTT butterflies; the latency-bound tail kernels keep the C++ form.

The 64-bit column accumulator lives in v[0:1] (an inline-asm operand cannot name the halves of a 64-bit register
pair, and v_mul_lo_u32 / v_and_b32 need the low half), declared as clobbers.

    python tools/gen_field_asm.py myzkp_amd/csrc/mzk_field_asm.h
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_constants import FIELDS, limbs, W

COL = "v[0:1]"
CLO = "v0"


class Asm:
    def __init__(self):
        self.outs, self.ins, self.lines = [], [], []

    def out(self, expr):
        self.outs.append(expr)
        return "%%%d" % (len(self.outs) - 1)

    def fix(self):
        self.nout = len(self.outs)

    def inp(self, cons, expr):
        self.ins.append((cons, expr))
        return "%%%d" % (self.nout + len(self.ins) - 1)

    def emit(self, s):
        self.lines.append(s)

    def render(self, indent="    "):
        body = "\n".join('%s    "%s\\n\\t"' % (indent, l) for l in self.lines)
        outs = ", ".join('"=&v"(%s)' % e for e in self.outs)
        ins = ", ".join('"%s"(%s)' % ce for ce in self.ins)
        return '%sasm(\n%s\n%s    : %s\n%s    : %s\n%s    : "v0", "v1", "vcc");' % (indent, body, indent, outs, indent, ins, indent)


def gen(kind, name, L, P):
    """kind: mul | sqr | mul_add2"""
    A = Asm()
    r = [A.out("r.l[%d]" % i) for i in range(L)]
    A.fix()
    a = [A.inp("v", "a.l[%d]" % i) for i in range(L)]
    if kind == "sqr":
        a2 = [A.inp("v", "a2[%d]" % i) for i in range(L - 1)]
    else:
        b = [A.inp("v", "b.l[%d]" % i) for i in range(L)]
    if kind == "mul_add2":
        c = [A.inp("v", "c.l[%d]" % i) for i in range(L)]
        d = [A.inp("v", "d.l[%d]" % i) for i in range(L)]
    p = [A.inp("s", "%sParams::P[%d]" % (name, i)) if P[i] else None for i in range(L)]
    n0 = A.inp("s", "%sParams::N0" % name)
    mask = "0x1fffffff"
    first = [True]

    def mad(x, y):
        A.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (COL, x, y, "0" if first[0] else COL))
        first[0] = False

    def products(k):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        if kind == "sqr":
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    mad(a2[i], a[k - i])
            if k % 2 == 0:
                mad(a[k // 2], a[k // 2])
        else:
            for i in range(lo, hi + 1):
                mad(a[i], b[k - i])
            if kind == "mul_add2":
                for i in range(lo, hi + 1):
                    mad(c[i], d[k - i])

    for k in range(L):
        products(k)
        for i in range(k):
            if P[k - i]:
                mad(r[i], p[k - i])
        A.emit("v_mul_lo_u32 %s, %s, %s" % (r[k], CLO, n0))
        A.emit("v_and_b32 %s, %s, %s" % (r[k], mask, r[k]))
        if P[0]:
            mad(r[k], p[0])
        A.emit("v_lshrrev_b64 %s, 29, %s" % (COL, COL))
    # r[j] doubles as m[j]: m[j] is last read in column j + L - 1, r[j] is written in column j + L
    for k in range(L, 2 * L - 1):
        products(k)
        for i in range(k - L + 1, L):
            if P[k - i]:
                mad(r[i], p[k - i])
        # the m of this slot (r[k-L]) was last used above (i = k-L+1 > k-L), so it can take the output limb now
        A.emit("v_and_b32 %s, %s, %s" % (r[k - L], mask, CLO))
        A.emit("v_lshrrev_b64 %s, 29, %s" % (COL, COL))
    A.emit("v_mov_b32 %s, %s" % (r[L - 1], CLO))
    T = "Fe<%sParams>" % name
    if kind == "mul":
        sig = "mul(const %s& a, const %s& b)" % (T, T)
        pre = ""
    elif kind == "sqr":
        sig = "sqr(const %s& a)" % T
        pre = "    u32 a2[%d];\n    for (int i = 0; i < %d; i++) a2[i] = a.l[i] << 1;\n" % (L - 1, L - 1)
    else:
        sig = "mul_add2(const %s& a, const %s& b, const %s& c, const %s& d)" % (T, T, T, T)
        pre = ""
    return "  static __device__ __forceinline__ %s %s {\n    %s r;\n%s%s\n    return r;\n  }\n" % (T, sig, T, pre, A.render())


def gen_sparse(name, L, P, signed):
    """fe_mul_sparse of mzk_field.h (p = PT 2^(29 (L-1)) + 1: M128) as one chain: 25 + 5 multiply-adds, the complements of the low
    limbs by v_bfi, no v_mul_lo.  signed: the limbs of `a` are i32 (the NTT's lazily accumulated butterflies): v_mad_i64_i32 and
    arithmetic shifts, signed top limb.  Same columns and sums as the C++ form (C = 0)."""
    assert P[0] == 1 and not any(P[1:L - 1])
    A = Asm()
    r = [A.out("r.l[%d]" % i) for i in range(L)]
    n = [A.out("n[%d]" % i) for i in range(L - 1)]
    A.fix()
    a = [A.inp("v", "a.l[%d]" % i) for i in range(L)]
    b = [A.inp("v", "b.l[%d]" % i) for i in range(L)]
    pt = A.inp("s", "%sParams::P[%d]" % (name, L - 1))
    msk = A.inp("s", "0x1fffffffu")
    c4 = A.inp("s", "(%sParams::P[%d] + 1u + (1u << 29))" % (name, L - 1))
    mask = "0x1fffffff"
    madop = "v_mad_i64_i32" if signed else "v_mad_u64_u32"
    shr = "v_ashrrev_i64" if signed else "v_lshrrev_b64"
    first = [True]

    def mad(x, y):
        A.emit("%s %s, vcc, %s, %s, %s" % (madop, COL, x, y, "0" if first[0] else COL))
        first[0] = False

    for k in range(L - 1):
        for i in range(k + 1):
            mad(a[i], b[k - i])
        A.emit("v_bfi_b32 %s, %s, 0, %s" % (n[k], CLO, msk))          # (~lo) & MASK
        A.emit("%s %s, 29, %s" % (shr, COL, COL))
    for k in range(L - 1, 2 * L - 1):
        for i in range(k - L + 1, L):
            mad(a[i], b[k - i])
        if k < 2 * L - 2:
            mad(n[k - L + 1], pt)
        if k == L - 1:
            mad("1", c4)                                                # the constants of both reduction steps
            A.emit("v_and_b32 %s, %s, %s" % (n[0], mask, CLO))
            A.emit("v_sub_u32 %s, 0x20000000, %s" % (n[0], n[0]))      # m = 2^29 - u0
            A.emit("%s %s, 29, %s" % (shr, COL, COL))
        else:
            if k == 2 * L - 2:
                mad(n[0], pt)
            A.emit("v_and_b32 %s, %s, %s" % (r[k - L], mask, CLO))
            if k < 2 * L - 2:
                A.emit("%s %s, 29, %s" % (shr, COL, COL))
            else:
                A.emit("v_alignbit_b32 %s, v1, v0, 29" % r[L - 1])
    T = "Fe<%sParams>" % name
    fn = "smul" if signed else "mul"
    doc = ("  // a: signed lazy limbs (i32, |a_i| < 2^31); b: limbs below 2^29; result: limbs 0..%d in [0, 2^29), signed top limb\n" % (L - 2)) if signed else ""
    return "%s  static __device__ __forceinline__ %s %s(const %s& a, const %s& b) {\n    %s r;\n    u32 n[%d];\n%s\n    return r;\n  }\n" % (
        doc, T, fn, T, T, T, L - 1, A.render())


def gen_shoup(name, L, p):
    """x * w mod p, w and wq = floor(w 2^(29 L) / p) as SCALAR operands (a wave-uniform twiddle): same columns as fe_shoup_mul."""
    A = Asm()
    r = [A.out("r.l[%d]" % i) for i in range(L)]
    q = [A.out("q[%d]" % i) for i in range(L)]
    A.fix()
    x = [A.inp("v", "x.l[%d]" % i) for i in range(L)]
    w = [A.inp("s", "w[%d]" % i) for i in range(L)]
    wq = [A.inp("s", "wq[%d]" % i) for i in range(L)]
    pc = [A.inp("s", "%sParams::RMP[%d]" % (name, i)) for i in range(L)]
    mask = "0x1fffffff"
    first = [True]

    def mad(a, b):
        A.emit("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (COL, a, b, "0" if first[0] else COL))
        first[0] = False

    for k in range(L - 2, 2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            mad(x[i], wq[k - i])
        if k >= L:
            A.emit("v_and_b32 %s, %s, %s" % (q[k - L], mask, CLO))
        A.emit("v_lshrrev_b64 %s, 29, %s" % (COL, COL))
    A.emit("v_mov_b32 %s, %s" % (q[L - 1], CLO))
    first[0] = True
    for k in range(L):
        for i in range(k + 1):
            mad(x[i], w[k - i])
        for i in range(k + 1):
            mad(q[i], pc[k - i])
        A.emit("v_and_b32 %s, %s, %s" % (r[k], mask, CLO))
        if k < L - 1:
            A.emit("v_lshrrev_b64 %s, 29, %s" % (COL, COL))
    T = "Fe<%sParams>" % name
    head = "  // w, wq: wave-uniform (the asm takes them as scalar registers)\n  static __device__ __forceinline__ %s shoup_mul(const %s& x, const u32* w, const u32* wq) {\n" % (T, T)
    return head + "    %s r;\n    u32 q[%d];\n%s\n    return r;\n  }\n" % (T, L, A.render())


def check_slot_reuse(L):
    # m[j] last read: column j + L - 1 ; r[j] written at column j + L (k >= L) -- and in column k >= L the reads of
    # m[i] are for i >= k - L + 1, i.e. never m[k-L].
    for k in range(L, 2 * L - 1):
        assert all(i != k - L for i in range(k - L + 1, L))


def main(out):
    S = []
    S.append("// GENERATED by tools/gen_field_asm.py -- do not edit.")
    S.append("// fe_mul / fe_sqr / fe_mul_add2 of mzk_field.h as single inline-asm blocks (device code of the throughput kernels")
    S.append("// only; same columns, same sums, bit-identical results).  See the generator for the why and the register use.")
    S.append("#pragma once")
    S.append('#include "mzk_field.h"')
    S.append("#if defined(__HIP_DEVICE_COMPILE__)")
    S.append("namespace mzk {")
    S.append("template <class P> struct FeAsm;")
    for name, (p, L, nw) in FIELDS.items():
        check_slot_reuse(L)
        P = limbs(p, L)
        S.append("template <> struct FeAsm<%sParams> {" % name)
        sparse = P[0] == 1 and not any(P[1:L - 1])
        for kind in ("mul", "sqr", "mul_add2"):
            if kind == "mul" and sparse:
                S.append(gen_sparse(name, L, P, False))
                S.append(gen_sparse(name, L, P, True))
            else:
                S.append(gen(kind, name, L, P))
        if name == "Fr":
            S.append(gen_shoup(name, L, p))
        S.append("};")
    S.append("}  // namespace mzk")
    S.append("#else   // host pass of hipcc / g++ host builds: the portable form (never executed for device work)")
    S.append("namespace mzk {")
    S.append("template <class P> struct FeAsm {")
    S.append("  static MZK_HD Fe<P> mul(const Fe<P>& a, const Fe<P>& b) { return fe_mul<P>(a, b); }")
    S.append("  static MZK_HD Fe<P> sqr(const Fe<P>& a) { return fe_sqr<P>(a); }")
    S.append("  static MZK_HD Fe<P> mul_add2(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) { return fe_mul_add2<P>(a, b, c, d); }")
    S.append("  static MZK_HD Fe<P> shoup_mul(const Fe<P>& x, const u32* w, const u32* wq) { return fe_shoup_mul<P>(x, w, wq); }")
    S.append("  static MZK_HD Fe<P> smul(const Fe<P>& a, const Fe<P>& b) { return fe_mul_sparse<P, true, 0>(a, b); }")
    S.append("};")
    S.append("}  // namespace mzk")
    S.append("#endif")
    open(out, "w").write("\n".join(S) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myzkp_amd", "csrc", "mzk_field_asm.h"))
