"""GPU parity for the G2 row: MSM over pk.powers_2 (kzg.rs:114) and the G2 half of
setup_kzg_with_full_g2 (kzg.rs:42-55), bit-exact vs the oracle (pinned to the reference's test_g2)."""
import random
import numpy as np
import pytest
import orc
from orc import P_FR as R, P_FQ as Q, G2_GEN as G2, G2_INF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


def test_reference_test_g2_relations_through_the_msm(mz):
    """bn128.rs:306-323 expressed as MSMs: 2g + g + g == 4g, 9g + 5g == 12g + 2g, r g == infinity"""
    g = orc.g2_to_arr([G2])
    one = lambda ks: mz.msm_g2(orc.to_limbs(ks, 4), np.repeat(g, len(ks), axis=0))
    assert one([2, 1, 1]) == one([4]) == orc.g2_mul(G2, 4)
    assert one([9, 5]) == one([12, 2])
    assert one([R - 1, 1]) == G2_INF
    assert mz.msm_g2(orc.to_limbs([R], 5)[:, :4], g) == orc.g2_mul(G2, R % (1 << 256) % R)    # unsanitised scalar == r -> canonicalised to 0


@pytest.mark.parametrize("n", [0, 1, 2, 5, 33, 200])
def test_msm_vs_literal_oracle(mz, n):
    rnd = random.Random(n)
    ks = [rnd.randrange(R) for _ in range(n)]
    pts = [orc.g2_mul(G2, rnd.randrange(1, R)) for _ in range(min(n, 12))]
    pts = [pts[i % len(pts)] for i in range(n)] if n else []
    s, p = orc.to_limbs(ks, 4), orc.g2_to_arr(pts)
    assert mz.msm_g2(s, p) == orc.g2_msm_ref(s, p)


def test_edge_batch(mz):
    P = orc.g2_mul(G2, 12345)
    neg = (P[0], ((-P[1][0]) % Q, (-P[1][1]) % Q))
    pts = [P, neg, G2_INF, P, P, G2, G2]
    ks = [7, 7, 99, 0, R - 1, 1, 1]                       # P and -P cancel, infinity point, zero scalar, r-1, repeated
    s, p = orc.to_limbs(ks, 4), orc.g2_to_arr(pts)
    assert mz.msm_g2(s, p) == orc.g2_msm_ref(s, p)
    assert mz.msm_g2(orc.to_limbs([5, 5], 4), orc.g2_to_arr([P, neg])) == G2_INF
    with pytest.raises(mz.MzkError) as e:                 # powers.len() < coef.len() (polynomial.rs:162)
        mz.msm_g2(orc.to_limbs([1, 2], 4), orc.g2_to_arr([P]))
    assert e.value.code == -5


def test_setup_full_g2_powers(mz):
    alpha = 0x1234567890abcdef1234567890abcdef1234567890abcdef % R
    got = mz.kzg_setup_g2(alpha, 40, G2)
    assert np.array_equal(got, orc.kzg_setup_g2_ref(alpha, 40))
    assert mz.array_to_g2_points(got[:1])[0] == G2
    assert not mz.kzg_setup_g2(0, 3, G2)[1:].any()         # alpha = 0: g2, infinity, infinity, ...


def test_batch_verify_shape_identity(mz):
    """the quantity batch_verify_kzg needs (kzg.rs:110-114): g2_z = z(alpha) g2 for z = prod (X - u_i), as an MSM
    over powers_2 -- checked against the trapdoor value [z(alpha)] g2"""
    alpha = 987654321987654321 % R
    us = [3, 5, 11, 1 << 100]
    z = [1]
    for u in us:                                           # from_monomials (polynomial.rs:202-212)
        z = [((z[i - 1] if i else 0) - u * (z[i] if i < len(z) else 0)) % R for i in range(len(z) + 1)]
    powers2 = mz.kzg_setup_g2(alpha, len(z) - 1, G2)
    za = sum(c * pow(alpha, i, R) for i, c in enumerate(z)) % R
    assert mz.msm_g2(orc.to_limbs(z, 4), powers2) == orc.g2_mul(G2, za)
