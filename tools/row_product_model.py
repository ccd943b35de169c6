#!/usr/bin/env python3
"""Integer model of the ROW-COOPERATIVE Montgomery product of myzkp_amd/csrc/mzk_row.h (one field element spread over the 16
lanes of a DPP row: lane j holds 29-bit limb j, lanes 9..15 zero), written lane by lane with the same data movement as the
device code -- row shifts with zero fill, 64-bit column accumulators, the three-piece split of a column -- so that every bound
the kernels rely on is an assertion here (tests/test_row_product_model.py runs it on random and extreme operands):

    columns     C_k = sum_i a_i b_(k-i)                         k = 0..16    (a replicated in the row, b distributed)
    low limbs   L'_k = lo(C_k) + mid(C_(k-1)) + hi(C_(k-2))     k < 9        == C mod R, limbs < 2^30.01, no ripple
    m           = normalise(low half of L' * N')   (N' = -p^-1 mod R, R = 2^261): == -C / p (mod R), m < 2.01 R
    C += m p    divisible by R; result = C / R, read from columns 9..17, plus the carry e of the low half, which is exact
                from ONE limb: the normalised low limbs n_0..n_8 sum to e R with e in {0..3}, and e = (n_8 + 4) >> 29.

No serial carry chain anywhere: a product is ~100 wave instructions deep instead of 214 on one lane."""
import random

W, L, LANES = 29, 9, 16
M29 = (1 << W) - 1
Rr = 1 << (W * L)
P_FQ = 21888242871839275222246405745257275088696311157297823662689037894645226208583
P_FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def limbs(x, n=L):
    return [(x >> (W * i)) & M29 for i in range(n - 1)] + [x >> (W * (n - 1))]


def value(l):
    return sum(v << (W * i) for i, v in enumerate(l))


def row_shr(v, n):      # lane j <- lane j - n, zero fill (DPP row_shr:n, bound_ctrl:0)
    return [v[j - n] if j - n >= 0 else 0 for j in range(LANES)]


def row_shl(v, n):      # lane j <- lane j + n, zero fill
    return [v[j + n] if j + n < LANES else 0 for j in range(LANES)]


def row_ror(v, n):      # lane j <- lane (j - n) mod 16
    return [v[(j - n) % LANES] for j in range(LANES)]


def split3(acc):
    for c in acc:
        assert 0 <= c < (1 << 64), "column accumulator overflows 64 bits"
    return [c & M29 for c in acc], [(c >> W) & M29 for c in acc], [c >> (2 * W) for c in acc]


def norm_low(acc):
    """n_j = lo_j + mid_(j-1) + hi_(j-2) in every lane (the caller masks the lanes it does not want)"""
    lo, mid, hi = split3(acc)
    m1, h2 = row_shr(mid, 1), row_shr(hi, 2)
    return [lo[j] + m1[j] + h2[j] for j in range(LANES)]


def mask9(v):
    return [v[j] if j < L else 0 for j in range(LANES)]


def row_mul(a, bD, p, a2=None, b2D=None):
    """a: 9 limbs (replicated operand), bD: 16 lanes (distributed operand).  Returns the distributed product a b / R mod p,
    limbs < 2^30.01, value < a b / R + 2.01 p.  With (a2, b2D): the fused pair (a b + a2 b2) / R, one reduction (mul2)."""
    pl = limbs(p)
    np_ = limbs((-pow(p, -1, Rr)) % Rr)
    assert all(x < (1 << 32) for x in a) and all(x < (1 << 32) for x in bD) and all(bD[j] == 0 for j in range(L, LANES))
    acc0 = [0] * LANES
    acc1 = [0] * LANES
    for (aa, bb) in ((a, bD),) + (((a2, b2D),) if a2 is not None else ()):
        for i in range(L):
            t = row_shr(bb, i)
            acc0 = [acc0[j] + aa[i] * t[j] for j in range(LANES)]
        t = row_ror(bb, 8)
        acc1 = [acc1[j] + aa[8] * t[j] for j in range(LANES)]     # lane 0: column 16; lane 1: 0; lanes >= 2: unused
    assert acc1[1] == 0
    # low limbs of the product, congruent mod R
    lp = mask9(norm_low(acc0))
    assert all(x < (1 << 30) + (1 << 7) for x in lp)
    macc = [0] * LANES
    for i in range(L):
        t = row_shr(lp, i)
        macc = [macc[j] + np_[i] * t[j] for j in range(LANES)]
    m = mask9(norm_low(macc))
    assert all(x < (1 << 30) + (1 << 7) for x in m)
    assert (value(m[:L]) - value(lp[:L]) * value(np_)) % Rr == 0
    for i in range(L):
        t = row_shr(m, i)
        acc0 = [acc0[j] + pl[i] * t[j] for j in range(LANES)]
    t = row_ror(m, 8)
    acc1 = [acc1[j] + pl[8] * t[j] for j in range(LANES)]
    assert acc1[1] == 0
    total = sum(acc0[j] << (W * j) for j in range(LANES)) + (acc1[0] << (W * 16))
    assert total % Rr == 0
    lo0, mid0, hi0 = split3(acc0)
    lo1, mid1, hi1 = split3(acc1)
    assert hi1[0] == 0
    n_all = norm_low(acc0)
    # the normalised low limbs n_0..n_8 form e R exactly, e small: recover e from n_8 alone
    vlow = sum(n_all[j] << (W * j) for j in range(L))
    assert vlow % Rr == 0 and vlow // Rr <= 3
    e_all = [(x + 4) >> W for x in n_all]
    assert e_all[8] == vlow // Rr, "carry of the low half is not exact"
    e_mv = row_shl(e_all, 8)                                   # lane 0 <- lane 8
    r = [0] * LANES
    a9, a8, a7, b7, b8 = row_shl(lo0, 9), row_shl(mid0, 8), row_shl(hi0, 7), row_shr(lo1, 7), row_shr(mid1, 8)
    for j in range(LANES):
        r[j] = a9[j] + a8[j] + a7[j] + b7[j] + b8[j] + (e_mv[j] if j == 0 else 0)
    r = mask9(r)
    assert value(r[:L]) == total // Rr
    assert all(x < (1 << 30) + (1 << 8) for x in r[:L - 1])
    return r


def check(a_val, b_val, p, a_limbs=None, b_limbs=None):
    a = a_limbs if a_limbs is not None else limbs(a_val)
    b = b_limbs if b_limbs is not None else limbs(b_val)
    r = row_mul(a, b + [0] * (LANES - L), p)
    va, vb, vr = value(a), value(b), value(r[:L])
    assert (vr * Rr - va * vb) % p == 0
    assert vr * Rr < va * vb + 2.02 * p * Rr
    return vr


def check2(a, b, a2, b2, p):
    r = row_mul(a, b + [0] * (LANES - L), p, a2, b2 + [0] * (LANES - L))
    vr = value(r[:L])
    tot = value(a) * value(b) + value(a2) * value(b2)
    assert (vr * Rr - tot) % p == 0 and vr * Rr < tot + 2.02 * p * Rr
    return vr


def kps(K, p):
    """K p with slack limbs, as tools/gen_constants.py emits FqRowParams::KPS"""
    c = limbs(K * p)
    return [c[j] + ((1 << 31) if j < 8 else 0) - (4 if j > 0 else 0) for j in range(L)]


def norm(x):
    """one parallel carry step: limbs < 2^32 -> limbs < 2^29 + 8 (top limb free)"""
    assert all(0 <= v < (1 << 32) for v in x)
    lo = [x[j] & M29 if j < 8 else x[j] for j in range(L)]
    c = [x[j] >> W if j < 8 else 0 for j in range(L)]
    return [lo[j] + (c[j - 1] if j else 0) for j in range(L)]


def sub(K, a, b, p):
    k = kps(K, p)
    assert all(k[j] >= b[j] for j in range(L)), "K p limb below the subtrahend's"
    return norm([a[j] + (k[j] - b[j]) for j in range(L)])


def lazy(x, rnd):
    """a lazily normalised representation of x: limbs up to 2^30 + 60, as a product output may have them"""
    l = limbs(x)
    for j in range(8):
        if l[j + 1] > 0 and rnd.random() < 0.7:
            l[j] += 1 << W
            l[j + 1] -= 1
    return l


def self_test(rounds=300, seed=1):
    rnd = random.Random(seed)
    for p in (P_FQ, P_FR):
        for _ in range(rounds):
            check(rnd.randrange(4 * p), rnd.randrange(4 * p), p)
        # extreme limbs: both operands lazily normalised (every limb at the bound the kernels allow), and all-ones canonical-width
        big = [(1 << 30) + 200] * 8 + [(1 << 25)]
        check(None, None, p, big, big)
        check(None, None, p, [M29] * 8 + [(1 << 24)], [(1 << 30) + 255] * 8 + [1 << 24])
        check(0, 0, p)
        check(p - 1, p - 1, p)
        check(1, Rr % p, p)
        # the fused pair of the last product level: three normalised operands (limbs <= 2^29 + 7) and one lazy product output
        n8 = [(1 << 29) + 7] * 8 + [1 << 24]
        check2(n8, n8, n8, [(1 << 30) + 60] * 8 + [1 << 24], p)
        for _ in range(rounds):
            x, y, z, w = (rnd.randrange(3 * p) for _ in range(4))
            check2(norm(lazy(x, rnd)), norm(lazy(y, rnd)), norm(lazy(z, rnd)), lazy(w, rnd), p)
        # a - b + K p limb-wise with the slack constants: no limb underflows, value exact
        for K, bmax in ((3, 2.5), (5, 4.5), (7, 6.5), (8, 7.5), (10, 9.5)):
            for _ in range(50):
                a, b = rnd.randrange(int(2.3 * p)), rnd.randrange(int(bmax * p))
                r = sub(K, lazy(a, rnd), lazy(b, rnd), p)
                assert value(r) == a - b + K * p and all(v < (1 << 29) + 8 for v in r[:8])
            r = sub(K, lazy(int(2.3 * p), rnd), lazy(int(bmax * p) - 1, rnd), p)
    return True


if __name__ == "__main__":
    print("row product model:", "ok" if self_test() else "FAILED")
