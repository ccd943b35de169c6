"""GPU parity: ntt::fast_coset_divide (ntt.rs:271-330) through the C ABI vs the oracle's literal restatement."""
import random
import numpy as np
import pytest
import orc
from orc import FR, M128

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


def _mul(p, a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % p
    return out


@pytest.mark.parametrize("fid", [M128, FR])
@pytest.mark.parametrize("dq,dr", [(2, 1), (5, 2), (3, 4), (9, 3), (40, 24), (100, 27), (5, 120), (300, 200)])
def test_exact_quotients(mz, fid, dq, dr):
    p, nl = orc.MOD[fid], orc.LIMBS[fid]
    rnd = random.Random(dq * 977 + dr)
    q = [rnd.randrange(p) for _ in range(dq)] + [rnd.randrange(1, p)]
    r = [rnd.randrange(p) for _ in range(dr)] + [rnd.randrange(1, p)]
    lhs = _mul(p, q, r)
    root, order = orc.root_of(fid, 11), 1 << 11
    offset = orc.M128_GEN if fid == M128 else 5
    a, b = orc.to_limbs(lhs + [0, 0, 0], nl), orc.to_limbs(r + [0], nl)        # trailing zeros: degree() trims them
    got = mz.fast_coset_divide(fid, a, b, offset, root, order)
    rc, want = orc.fast_coset_divide_ref(fid, a, b, offset, root, order)
    assert rc == 0 and np.array_equal(got, want)
    assert orc.from_limbs(got) == q


@pytest.mark.parametrize("fid,ll,lr,lg", [(M128, 50, 20, 9), (FR, 77, 13, 8), (M128, 1 << 12, 1 << 10, 13), (M128, 9, 8, 6), (M128, 7, 3, 6)])
def test_inexact_matches_oracle_recipe(mz, fid, ll, lr, lg):
    """rhs does not divide lhs (and rhs may vanish on coset points): the recipe's own result, incl. inverse(0) = 0"""
    p, nl = orc.MOD[fid], orc.LIMBS[fid]
    lhs = orc.synth_vector(fid, 800 + ll, ll)
    rhs = orc.synth_vector(fid, 801 + lr, lr)
    root, order = orc.root_of(fid, lg), 1 << lg
    offset = orc.M128_GEN if fid == M128 else 7
    got = mz.fast_coset_divide(fid, lhs, rhs, offset, root, order)
    rc, want = orc.fast_coset_divide_ref(fid, lhs, rhs, offset, root, order)
    assert rc == 0 and np.array_equal(got, want)


def test_divisor_vanishing_on_the_coset(mz):
    """rhs = X - offset has a root ON the evaluation coset: that codeword entry divides by zero -> el * 0 (field.rs:209-232)"""
    p = orc.MOD[M128]
    offset = orc.M128_GEN
    lhs = orc.synth_vector(M128, 5, 40)
    rhs = orc.to_limbs([(p - offset) % p, 1], 2)
    root, order = orc.root_of(M128, 6), 64
    got = mz.fast_coset_divide(M128, lhs, rhs, offset, root, order)
    rc, want = orc.fast_coset_divide_ref(M128, lhs, rhs, offset, root, order)
    assert rc == 0 and np.array_equal(got, want)


def test_assertions(mz):
    z = orc.to_limbs([0, 0, 0], 2)
    a = orc.to_limbs(list(range(1, 12)), 2)
    b = orc.to_limbs([1, 2, 3], 2)
    root = orc.root_of(M128, 6)
    for args, code in (((a, z, 3, root, 64), -1), ((b, a, 3, root, 64), -5), ((z, b, 3, root, 64), -5), ((a, b, 3, root, 32), -3),
                       ((a, b, 3, root, 128), -4)):
        with pytest.raises(mz.MzkError) as e:
            mz.fast_coset_divide(M128, *args)
        assert e.value.code == code
        assert orc.fast_coset_divide_ref(M128, *args)[0] == code
    big = orc.synth_vector(M128, 1, 100)
    with pytest.raises(mz.MzkError) as e:                      # more coefficients than the root's order
        mz.fast_coset_divide(M128, big, b, 3, root, 64)
    assert e.value.code == orc.fast_coset_divide_ref(M128, big, b, 3, root, 64)[0] == -2
