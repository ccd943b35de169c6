"""CPU: the oracle's G2 group law over Fq2 pinned by the reference's own test_g2 (bn128.rs:306-323) and by an
independent pure-Python Fq2 / affine restatement."""
import random
import numpy as np
import orc
from orc import P_FQ as Q, P_FR as R, G2_GEN as G2, G2_INF


def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * n % Q, -a[1] * n % Q)


def py_add(P, S):
    if P == G2_INF: return S
    if S == G2_INF: return P
    (x1, y1), (x2, y2) = P, S
    if x1 == x2:
        if y1 != y2: return G2_INF
        lam = f2mul(f2mul((3, 0), f2mul(x1, x1)), f2inv(f2add(y1, y1)))
    else:
        lam = f2mul(f2sub(y2, y1), f2inv(f2sub(x2, x1)))
    x3 = f2sub(f2sub(f2mul(lam, lam), x1), x2)
    return (x3, f2sub(f2mul(lam, f2sub(x1, x3)), y1))


def py_mul(P, k):
    acc, cur = G2_INF, P
    while k:
        if k & 1: acc = py_add(acc, cur)
        cur = py_add(cur, cur)
        k >>= 1
    return acc


def test_reference_test_g2():
    """bn128.rs:306-323"""
    assert orc.g2_on_curve(G2)                                              # y^2 - x^3 == b2 = 3 / (9 + u)
    g = G2
    assert orc.g2_add(orc.g2_add(orc.g2_mul(g, 2), g), g) == orc.g2_mul(orc.g2_mul(g, 2), 2)
    assert orc.g2_add(orc.g2_mul(g, 9), orc.g2_mul(g, 5)) == orc.g2_add(orc.g2_mul(g, 12), orc.g2_mul(g, 2))
    assert orc.g2_mul(g, R) == G2_INF


def test_against_python_restatement():
    rnd = random.Random(2)
    for _ in range(6):
        a, b = rnd.randrange(1, R), rnd.randrange(1, R)
        A, B = orc.g2_mul(G2, a), orc.g2_mul(G2, b)
        assert A == py_mul(G2, a) and orc.g2_on_curve(A)
        assert orc.g2_add(A, B) == py_add(A, B) == orc.g2_mul(G2, (a + b) % R)
    P = orc.g2_mul(G2, 77)
    neg = (P[0], ((-P[1][0]) % Q, (-P[1][1]) % Q))
    assert orc.g2_add(P, neg) == G2_INF and orc.g2_add(P, P) == py_add(P, P) == orc.g2_mul(G2, 154)
    assert orc.g2_add(G2_INF, P) == P and orc.g2_add(P, G2_INF) == P and orc.g2_mul(P, 0) == G2_INF


def test_msm_and_setup_literal():
    rnd = random.Random(3)
    ks = [rnd.randrange(R) for _ in range(5)]
    pts = [orc.g2_mul(G2, rnd.randrange(1, R)) for _ in range(5)]
    want = G2_INF
    for k, P in zip(ks, pts):
        want = py_add(want, py_mul(P, k))
    assert orc.g2_msm_ref(orc.to_limbs(ks, 4), orc.g2_to_arr(pts)) == want
    srs = orc.kzg_setup_g2_ref(7, 4)
    assert orc.arr_to_g2(srs) == [py_mul(G2, pow(7, i, R)) for i in range(5)]
