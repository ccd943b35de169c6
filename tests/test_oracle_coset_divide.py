"""CPU: the oracle's fast_coset_divide (ntt.rs:271-330) against a pure-Python restatement and against true
polynomial division (when rhs divides lhs the coset trick returns the exact quotient)."""
import random
import numpy as np
import pytest
import orc
from orc import FR, M128


def py_trim(c):
    while c and c[-1] == 0:
        c = c[:-1]
    return c


def py_ntt(p, root, v):        # ntt.rs:7-48 (recursive, even/odd split)
    n = len(v)
    if n <= 1:
        return list(v)
    half = n // 2
    ev = py_ntt(p, root * root % p, v[0::2])
    od = py_ntt(p, root * root % p, v[1::2])
    return [(ev[i % half] + pow(root, i, p) * od[i % half]) % p for i in range(n)]


def py_divrem_quo(p, a, b):    # polynomial.rs:371-405
    a, b = py_trim(list(a)), py_trim(list(b))
    q = [0] * (len(a) - len(b) + 1)
    li = pow(b[-1], -1, p)
    while len(a) >= len(b):
        lead = a[-1] * li % p
        dd = len(a) - len(b)
        q[dd] = lead
        for i in range(len(b)):
            a[dd + i] = (a[dd + i] - lead * b[i]) % p
        a = py_trim(a)
    return py_trim(q)


def py_coset_divide(p, lhs, rhs, offset, root, order):     # ntt.rs:271-330
    dl, dr = len(py_trim(list(lhs))) - 1, len(py_trim(list(rhs))) - 1
    if dl < 8:
        return py_divrem_quo(p, lhs, rhs)
    while dl < order // 2:
        root = root * root % p
        order //= 2
    a = [lhs[i] * pow(offset, i, p) % p for i in range(dl + 1)] + [0] * (order - dl - 1)
    b = [rhs[i] * pow(offset, i, p) % p for i in range(dr + 1)] + [0] * (order - dr - 1)
    ea, eb = py_ntt(p, root, a), py_ntt(p, root, b)
    qc = [x * (pow(y, -1, p) if y else 0) % p for x, y in zip(ea, eb)]
    ninv, rinv, oinv = pow(order, -1, p), pow(root, -1, p), pow(offset, -1, p)
    sq = [x * ninv % p for x in py_ntt(p, rinv, qc)][: dl - dr + 1]
    return [c * pow(oinv, i, p) % p for i, c in enumerate(sq)]


def py_mul(p, a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % p
    return out


@pytest.mark.parametrize("fid", [M128, FR])
@pytest.mark.parametrize("dq,dr", [(3, 2), (6, 1), (9, 3), (40, 24), (100, 27), (5, 120)])
def test_exact_division_and_python_restatement(fid, dq, dr):
    p, nl = orc.MOD[fid], orc.LIMBS[fid]
    rnd = random.Random(dq * 1000 + dr)
    q = [rnd.randrange(p) for _ in range(dq)] + [rnd.randrange(1, p)]
    r = [rnd.randrange(p) for _ in range(dr)] + [rnd.randrange(1, p)]
    lhs = py_mul(p, q, r)
    lg = 10
    root, order = orc.root_of(fid, lg), 1 << lg
    offset = orc.M128_GEN if fid == M128 else 5
    rc, got = orc.fast_coset_divide_ref(fid, orc.to_limbs(lhs + [0, 0], nl), orc.to_limbs(r + [0], nl), offset, root, order)
    assert rc == 0
    assert orc.from_limbs(got) == py_coset_divide(p, lhs, r, offset, root, order)
    assert orc.from_limbs(got) == q                     # rhs | lhs: the coset trick is exact


def test_inexact_division_matches_python_restatement():
    """rhs does not divide lhs: the function returns what its own recipe computes, not the true quotient"""
    p = orc.MOD[M128]
    rnd = random.Random(5)
    lhs = [rnd.randrange(p) for _ in range(50)]
    rhs = [rnd.randrange(p) for _ in range(20)]
    root, order = orc.root_of(M128, 9), 512
    rc, got = orc.fast_coset_divide_ref(M128, orc.to_limbs(lhs, 2), orc.to_limbs(rhs, 2), orc.M128_GEN, root, order)
    assert rc == 0 and orc.from_limbs(got) == py_coset_divide(p, lhs, rhs, orc.M128_GEN, root, order)
    assert len(got) == 50 - 20 + 1


def test_contract_violations():
    z = orc.to_limbs([0, 0, 0], 2)
    a = orc.to_limbs(list(range(1, 12)), 2)
    b = orc.to_limbs([1, 2, 3], 2)
    root = orc.root_of(M128, 6)
    assert orc.fast_coset_divide_ref(M128, a, z, 3, root, 64)[0] == -1          # !rhs.is_zero()
    assert orc.fast_coset_divide_ref(M128, b, a, 3, root, 64)[0] == -5          # rhs.degree() < lhs.degree()
    assert orc.fast_coset_divide_ref(M128, z, b, 3, root, 64)[0] == -5          # zero lhs: degree -1
    assert orc.fast_coset_divide_ref(M128, a, b, 3, root, 32)[0] == -3          # root^order != 1
    assert orc.fast_coset_divide_ref(M128, a, b, 3, root, 128)[0] == -4         # root^(order/2) == 1
