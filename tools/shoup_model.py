#!/usr/bin/env python3
"""Integer model of the precomputed-quotient (Shoup) product of the NTT's wave-uniform twiddles (FeAsm<FrParams>::shoup_mul,
fe_shoup_mul in mzk_field.h): x * w mod p for a constant w with wq = floor(w 2^261 / p), x in the lazy form the butterflies
hand to a product (limbs up to 3 * 2^30, value below 2^261), every column a 64-bit accumulator as on the device.

    q = columns 9 .. 17 of x * wq (columns 7, 8 computed as guards, lower ones dropped)
    r = low 9 columns of x * w + q * (2^261 - p)

Checked: no column overflows 64 bits, r == x w (mod p), r < 4 p, limbs of r below 2^29.
    python tools/shoup_model.py [cases]
"""
import random, sys

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
W, L = 29, 9
MASK = (1 << W) - 1
BETA = 1 << (W * L)
PC = [((BETA - P) >> (W * i)) & MASK for i in range(L)]


def limbs(v):
    return [(v >> (W * i)) & MASK for i in range(L)]


def value(l):
    return sum(v << (W * i) for i, v in enumerate(l))


def shoup_mul(x, w, wq):
    """x: lazy limbs; w, wq: normalised limbs.  Returns (r limbs, worst column)."""
    worst = 0
    col = 0
    q = [0] * L
    for k in range(7, 2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            col += x[i] * wq[k - i]
        worst = max(worst, col)
        if k >= L:
            q[k - L] = col & MASK
        col >>= W
    q[L - 1] = col & 0xffffffff
    assert col < 1 << 32
    r = [0] * L
    col = 0
    for k in range(L):
        for i in range(k + 1):
            col += x[i] * w[k - i]
        for i in range(k + 1):
            col += q[i] * PC[k - i]
        worst = max(worst, col)
        r[k] = col & MASK
        col >>= W
    return r, worst, q


def lazy_operand(rng, kind):
    lim = {0: MASK, 1: 3 << 30, 2: (1 << 30) + (1 << 29)}[kind % 3]
    while True:
        x = [rng.randrange(lim + 1) for _ in range(L)]
        if kind >= 3:
            x = [lim] * L
        x[L - 1] = rng.randrange(1 << 27)          # the value stays below 2^261 (the butterflies keep it below ~80 p)
        if kind >= 3:
            x[L - 1] = (1 << 27) - 1
        if value(x) < BETA:
            return x


def main(cases):
    rng = random.Random(11)
    worst = 0
    rmax = 0
    for c in range(cases):
        wv = rng.randrange(P) if c % 7 else [0, 1, P - 1, P // 2, 2, P - 2, (1 << 253)][(c // 7) % 7]
        w, wq = limbs(wv), limbs(wv * BETA // P)
        x = lazy_operand(rng, c % 4 if c % 50 else 3)
        r, wc, q = shoup_mul(x, w, wq)
        worst = max(worst, wc)
        assert wc < 1 << 64, "column overflow"
        rv = value(r)
        assert rv % P == value(x) * wv % P, "wrong residue"
        assert rv < 4 * P, ("r >= 4p", rv / P)
        assert all(v <= MASK for v in r)
        rmax = max(rmax, rv / P)
    print("%d cases: worst column 2^%.3f, largest r = %.3f p" % (cases, __import__("math").log2(worst), rmax))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20000)
