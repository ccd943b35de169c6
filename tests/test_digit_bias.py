"""Signed window digits without the serial carry walk (k_small_accumulate_scan, mzk_msm.hip): with
t = k + sum_w (2^(c-1) - 1) 2^(c w), digit_w = window_w(t) - (2^(c-1) - 1) must equal the carry-walking recoding of
walk_digits (digits in (-2^(c-1), 2^(c-1)], raw > half borrows from the next window) for every scalar below r."""
import random

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def walk_digits(k, c):
    nwin = 254 // c + 1
    half = 1 << (c - 1)
    out, carry = [], 0
    for w in range(nwin):
        raw = ((k >> (c * w)) & ((1 << c) - 1)) + carry
        carry = 0
        d = raw
        if raw > half:
            d = raw - (1 << c)
            carry = 1
        out.append(d)
    assert carry == 0
    return out


def biased_digits(k, c):
    nwin = 254 // c + 1
    half = 1 << (c - 1)
    bias = sum((half - 1) << (c * w) for w in range(nwin))
    t = k + bias
    assert t < 1 << 288                                   # nine 32-bit words in the kernel
    return [((t >> (c * w)) & ((1 << c) - 1)) - (half - 1) for w in range(nwin)]


def test_biased_windows_equal_the_carry_walk():
    rng = random.Random(5)
    for c in (8, 10, 11, 12, 13, 16, 17):           # the widths with a compile-time walker (mzk_msm.hip)
        half = 1 << (c - 1)
        special = [0, 1, R - 1, R - 2, half, half + 1, half - 1, (1 << 254) - 1 if (1 << 254) - 1 < R else R - 1]
        special += [sum(half << (c * w) for w in range(254 // c)) % R, sum((half + 1) << (c * w) for w in range(254 // c)) % R]
        for k in special + [rng.randrange(R) for _ in range(3000)]:
            a, b = walk_digits(k, c), biased_digits(k, c)
            assert a == b, (c, hex(k))
            assert sum(d << (c * w) for w, d in enumerate(a)) == k
