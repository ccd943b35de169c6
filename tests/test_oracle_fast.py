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


def test_msm_fast_sliced_over_many_threads_equals_one_thread_and_trapdoor():
    """orc_msm_fast splits every window over up to 16 slices of the pairs when it has more threads than windows (the GPU box's
    256 cores): same point as one thread, and on SRS points [alpha^i]G the closed form [f(alpha)]G (polynomial.rs:156-165)."""
    n, alpha = 1 << 13, 0x1234567
    s = orc.synth_vector(FR, 4001, n)
    powers = orc.to_limbs([pow(alpha, i, orc.P_FR) for i in range(n)], 4)
    p = orc.fixed_base_batch((1, 2), powers)    # [alpha^i] G (the literal setup_kzg restatement takes a minute at this size)
    p[77] = 0                                   # a point at infinity in the middle of a slice
    one = orc.msm_fast(s, p, threads=1)
    for t in (52, 80, 200):                     # 26 windows at c = 10: 2, 3 and (capped by the slice-size rule) 4 slices each
        assert orc.msm_fast(s, p, threads=t) == one, t
    s[77] = 0
    assert orc.msm_fast(s, p, threads=80) == orc.ec_mul(0, (1, 2), orc.poly_eval(FR, s, alpha))


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
