"""GPU parity for the "next" rows: FRI split-and-fold, batch_open_kzg, prove_degree_bound."""
import numpy as np
import pytest
import orc
from orc import FR, M128, P_FR, I

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


def test_fri_fold_golden_and_oracle(mz):
    for c in orc.golden("fri_vectors.json")["cases"]:
        f = {"Fr": FR, "M128": M128}[c["field"]]
        out = mz.fri_fold(f, orc.to_limbs(I(c["input"]), orc.LIMBS[f]), int(c["alpha"]), int(c["offset"]), int(c["omega"]))
        assert orc.from_limbs(out) == I(c["output"])
    for fid, lg in ((M128, 12), (FR, 11), (M128, 1)):
        n = 1 << lg
        cw = orc.synth_vector(fid, 70 + lg, n)
        alpha = orc.from_limbs(orc.synth_vector(fid, 71, 1))[0]
        off = orc.M128_GEN if fid == M128 else 5
        om = orc.root_of(fid, lg)
        assert np.array_equal(mz.fri_fold(fid, cw, alpha, off, om), orc.fri_fold_ref(fid, cw, alpha, off, om))
    assert mz.fri_fold(M128, orc.to_limbs([5], 2), 1, 2, 1).shape[0] == 0      # len 1 -> empty (len / 2 == 0)


def test_fri_commit_chain_matches_reference_recurrence(mz):
    # three rounds of FRI::commit's loop (fri.rs:154-196): fold, omega <- omega^2, offset <- offset^2
    n = 1 << 14
    cw = orc.synth_vector(M128, 9, n)
    omega, offset = orc.m128_root(14), orc.M128_GEN
    cur_gpu, cur_cpu = cw, cw
    for r in range(3):
        alpha = orc.from_limbs(orc.synth_vector(M128, 100 + r, 1))[0]
        cur_gpu = mz.fri_fold(M128, cur_gpu, alpha, offset, omega)
        cur_cpu = orc.fri_fold_ref(M128, cur_cpu, alpha, offset, omega)
        assert np.array_equal(cur_gpu, cur_cpu)
        omega, offset = omega * omega % orc.P_M128, offset * offset % orc.P_M128


def test_batch_open_and_degree_bound(mz):
    g = orc.golden("curve_vectors.json")
    for c in g["kzg_batch_open"]:
        srs = orc.pts_to_arr([tuple(x) for x in I(c["srs"])])
        ys, w = mz.kzg_batch_open(orc.to_limbs(I(c["coef"]), 4), I(c["us"]), srs)
        assert ys == I(c["ys"]) and list(w) == I(c["w"])
    for c in g["kzg_degree_bound"]:
        srs = orc.pts_to_arr([tuple(x) for x in I(c["srs"])])
        assert list(mz.kzg_prove_degree_bound(orc.to_limbs(I(c["coef"]), 4), srs, c["d"])) == I(c["out"])
    srs = orc.pts_to_arr([tuple(x) for x in I(g["kzg_degree_bound"][0]["srs"])])
    coef = orc.to_limbs(I(g["kzg_degree_bound"][0]["coef"]), 4)
    for d in (3, 17):
        with pytest.raises(mz.MzkError) as e:
            mz.kzg_prove_degree_bound(coef, srs, d)
        assert e.value.code == -5
    # larger: n = 2^12, 4 points, vs the oracle's literal restatement and the trapdoor identity
    n, k = 1 << 12, 4
    alpha = 0xabcdef12345
    srs = mz.kzg_setup_g1(alpha, n - 1)
    f = orc.synth_vector(FR, 31, n)
    us = orc.from_limbs(orc.synth_vector(FR, 32, k))
    ys, w = mz.kzg_batch_open(f, us, srs)
    assert ys == [orc.poly_eval(FR, f, u) for u in us]
    # q(alpha) = (f(alpha) - I(alpha)) / Z(alpha), I = Lagrange interpolant of (us, ys)
    fa = orc.poly_eval(FR, f, alpha)
    Ia = 0
    for j in range(k):
        num, den = 1, 1
        for i in range(k):
            if i != j:
                num = num * (alpha - us[i]) % P_FR
                den = den * (us[j] - us[i]) % P_FR
        Ia = (Ia + ys[j] * num * pow(den, -1, P_FR)) % P_FR
    Za = 1
    for u in us:
        Za = Za * (alpha - u) % P_FR
    assert w == orc.ec_mul(0, (1, 2), (fa - Ia) * pow(Za, -1, P_FR) % P_FR)
    ys_o, w_o = orc.kzg_batch_open_ref(f[:64], us, srs[:64])
    ys_g, w_g = mz.kzg_batch_open(f[:64], us, srs[:64])
    assert ys_g == ys_o and w_g == w_o


@pytest.mark.parametrize("fid,lg", [(M128, 17), (FR, 17), (M128, 21), (FR, 21)])
def test_fri_fold_every_walk_length(mz, fid, lg):
    """The fold kernel walks 1, 4 or 16 consecutive positions per lane depending on the codeword length (short late rounds vs
    long early ones): 2^17 takes the 4-step walk, 2^21 the 16-step one (2^14 and below, above: one step).  The whole vector
    against the oracle at 2^17, 64 sampled positions against the formula of fri.rs:182-193 at 2^21."""
    n, h = 1 << lg, 1 << (lg - 1)
    p = orc.MOD[fid]
    cw = orc.synth_vector(fid, 170 + lg, n)
    alpha = orc.from_limbs(orc.synth_vector(fid, 171, 1))[0]
    off = orc.M128_GEN if fid == M128 else 7
    om = orc.root_of(fid, lg)
    got = mz.fri_fold(fid, cw, alpha, off, om)
    assert got.shape[0] == h
    if lg <= 17:
        assert np.array_equal(got, orc.fri_fold_ref(fid, cw, alpha, off, om))
    rng = np.random.default_rng(lg)
    half = pow(2, -1, p)
    for i in [0, 1, 3, 4, 15, 16, 17, h - 1] + [int(k) for k in rng.integers(0, h, 56)]:
        a, b = orc.from_limbs(cw[i:i + 1])[0], orc.from_limbs(cw[h + i:h + i + 1])[0]
        q = alpha * pow(off * pow(om, i, p) % p, -1, p) % p
        assert orc.from_limbs(got[i:i + 1])[0] == half * ((1 + q) * a + (1 - q) * b) % p, i
