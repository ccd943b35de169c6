"""The oracle's fast CPU layer (iterative NTT, Pippenger) against its literal layer.  CPU only."""
import numpy as np
import orc
from orc import FR, M128


def test_ntt_fast_matches_ref():
    for fid in (FR, M128):
        for lg in (1, 3, 7, 11):
            n = 1 << lg
            v = orc.synth_vector(fid, 1234 + lg, n)
            w = orc.root_of(fid, lg)
            for inv in (False, True):
                rc0, a = (orc.intt_ref if inv else orc.ntt_ref)(fid, w, v)
                rc1, b = orc.ntt_fast(fid, w, v, inverse=inv, threads=4)
                assert rc0 == 0 and rc1 == 0 and np.array_equal(a, b)


def test_msm_fast_matches_ref():
    for n in (1, 7, 100, 300):
        s = orc.synth_vector(FR, 99 + n, n)
        p = orc.synth_points(7 + n, n)
        for i in range(n):
            assert orc.lib().orc_g1_on_curve(orc.ptr(p[i:i + 1])) == 1
        assert orc.msm_fast(s, p) == orc.msm_ref(s, p)


def test_fixed_base_batch_matches_ec_mul():
    s = orc.synth_vector(FR, 5, 16)
    out = orc.arr_to_pts(orc.fixed_base_batch((1, 2), s))
    ks = orc.from_limbs(s)
    for k, P in zip(ks, out):
        assert orc.ec_mul(0, (1, 2), k) == P


def test_synth_is_deterministic_and_canonical():
    a = orc.synth_vector(FR, 42, 64, threads=1)
    b = orc.synth_vector(FR, 42, 64, threads=8)
    assert np.array_equal(a, b)
    assert all(x < orc.P_FR for x in orc.from_limbs(a))
    assert all(x < orc.P_M128 for x in orc.from_limbs(orc.synth_vector(M128, 42, 64)))
