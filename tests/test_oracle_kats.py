"""Pin the CPU oracle: every known-answer test the reference holds for the MSM/NTT path
 plus the committed golden vectors from the independent Python transcription
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import numpy as np
import pytest
import orc
from orc import FR, FQ, M128, F631, F17, F31, P_FR, P_FQ, P_M128, I


def test_field_rs_small_kats():
    # field.rs:491-497: 7^-1 = 5 (mod 17)
    assert orc.field_op("inv", F17, 7) == 5
    # field.rs:443-489 small add/sub/mul mod 17
    assert orc.field_op("add", F17, 10, 15) == 8
    assert orc.field_op("sub", F17, 5, 10) == 12
    assert orc.field_op("mul", F17, 5, 10) == 16
    assert orc.field_pow(F17, 3, 4) == 13  # 3^4 mod 17
    # field.rs:499-504 / F6: -fe(10) compares equal to 7 after sanitize
    assert orc.field_op("neg", F17, 10) == 7
    # field.rs:544-550: -23 == 8 (mod 31)
    assert orc.field_op("neg", F31, 23) == 8


def test_test_fr_cu_kats():
    # cuda/test_fr.cu:16-42 limb-level identities over Fr
    pm2 = int("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593efffffff", 16)
    pm12 = int("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593effffff5", 16)
    assert pm2 == P_FR - 2 and pm12 == P_FR - 12
    assert orc.field_op("add", FR, 5, 7) == 12
    assert orc.field_op("sub", FR, 7, 5) == 2
    assert orc.field_op("sub", FR, 5, 7) == pm2
    assert orc.field_op("mul", FR, 5, 7) == 35
    assert orc.field_op("add", FR, 1, P_FR - 1) == 0
    assert orc.field_op("sub", FR, 0, 12) == pm12
    assert orc.field_op("mul", FR, pm2, pm12) == 24
    assert orc.field_op("add", FR, orc.field_op("mul", FR, pm12, 2), 24) == 0
    # cuda/kernels/field.hpp:9-31 constants
    k = orc.golden("field_kats.json")["reference_kats"]["field_hpp_9_31"]
    assert int(k["R2_mod_N"]) == pow(2, 512, P_FR)


def test_golden_field_cases():
    g = orc.golden("field_kats.json")
    fid = {"Fr": FR, "Fq": FQ, "M128": M128}
    for c in g["cases"]:
        f = fid[c["field"]]
        a, b = int(c["a"]), int(c["b"])
        assert orc.field_op("add", f, a, b) == int(c["add"])
        assert orc.field_op("sub", f, a, b) == int(c["sub"])
        assert orc.field_op("mul", f, a, b) == int(c["mul"])
        assert orc.field_op("neg", f, a) == int(c["neg"])
        assert orc.field_op("inv", f, a) == int(c["inv"])
        assert orc.field_pow(f, a, int(c["pow_e"])) == int(c["pow"])


def test_rescue_prime_m128_kat():
    # zkstark/rescueprime.rs:606-620, run on the oracle's M128 arithmetic
    par = orc.golden("rescue_prime_m128.json")
    m, N, alpha, ainv = par["m"], par["n"], int(par["alpha"]), int(par["alphainv"])
    mds, rc = I(par["mds"]), I(par["round_constants"])
    add = lambda a, b: orc.field_op("add", M128, a, b)
    mul = lambda a, b: orc.field_op("mul", M128, a, b)
    for kat in par["kats"]:
        state = [int(kat["input"])] + [0] * (m - 1)
        for r in range(N):
            for half, e in ((0, alpha), (1, ainv)):
                state = [orc.field_pow(M128, s, e) for s in state]
                t = [0] * m
                for i in range(m):
                    for j in range(m):
                        t[i] = add(t[i], mul(mds[i][j], state[j]))
                state = [add(t[i], rc[2 * r * m + half * m + i]) for i in range(m)]
        assert state[0] == int(kat["hash"])


def test_m128_root_of_unity():
    # fri.rs:436-438: generator has order exactly 2^119; get_nth_root_of_m128 squares it down
    g = orc.m128_root(119)
    assert g == orc.M128_GEN
    assert orc.field_pow(M128, g, 1 << 119) == 1 and orc.field_pow(M128, g, 1 << 118) != 1
    assert orc.m128_root(3) == 131076302407280330469229082343774091404
    for lg in (1, 8, 20, 24):
        w = orc.m128_root(lg)
        assert orc.field_pow(M128, w, 1 << lg) == 1 and orc.field_pow(M128, w, 1 << (lg - 1)) != 1
    assert orc.lib().orc_m128_nth_root(120, orc.ptr(np.zeros(2, dtype=np.uint64))) == -2


def test_ntt_rs_test_ntt():
    # ntt.rs:346-374: n = 256 over M128, coefficients 1..256: ntt == eval_domain; intt(ntt) == id
    n = 256
    w = orc.m128_root(8)
    coef = orc.to_limbs(list(range(1, n + 1)), 2)
    rc, vals = orc.ntt_ref(M128, w, coef)
    assert rc == 0
    v = orc.from_limbs(vals)
    for i in (0, 1, 2, 17, 128, 255):
        assert v[i] == orc.poly_eval(M128, coef, orc.field_pow(M128, w, i))
    rc, back = orc.intt_ref(M128, w, vals)
    assert rc == 0 and orc.from_limbs(back) == list(range(1, n + 1))
    h = hashlib.sha256(b"".join(int(x).to_bytes(16, "little") for x in v)).hexdigest()
    assert h == "8f7102fed15fa3c89551253cd93779f981c69431c46075b5507178c490920e19"


def test_ntt_assertions():
    coef = orc.to_limbs([1, 2, 3, 4, 5, 6, 7, 8], 2)
    assert orc.ntt_ref(M128, orc.m128_root(4), coef)[0] == -3      # order-16 root: root^8 != 1 (ntt.rs:15-18)
    assert orc.ntt_ref(M128, orc.m128_root(2), coef)[0] == -4      # order-4 root: root^4 == 1 (ntt.rs:19-22)
    assert orc.ntt_ref(M128, orc.m128_root(2), coef[:4])[0] == 0
    assert orc.ntt_ref(M128, orc.m128_root(1), coef[:4])[0] == -4  # order-2 root is not primitive for n = 4
    assert orc.ntt_ref(M128, orc.m128_root(3), coef[:6])[0] == -2  # not a power of two


def _fid(name):
    return {"Fr": FR, "M128": M128}[name]


def test_golden_ntt_vectors():
    g = orc.golden("ntt_vectors.json")
    seen = set()
    for c in g["cases"]:
        f = _fid(c["field"])
        nl = orc.LIMBS[f]
        kind = c["kind"]
        seen.add(kind)
        if kind in ("ntt", "intt"):
            inp = orc.to_limbs(I(c["input"]), nl)
            fn = orc.ntt_ref if kind == "ntt" else orc.intt_ref
            rc, out = fn(f, int(c["root"]), inp)
            assert rc == 0 and orc.from_limbs(out) == I(c["output"]), (kind, c["field"], len(c["input"]))
            rc, out = orc.ntt_fast(f, int(c["root"]), inp, inverse=(kind == "intt"))
            assert rc == 0 and orc.from_limbs(out) == I(c["output"])
        elif kind == "coset":
            rc, out = orc.coset_ref(f, orc.to_limbs(I(c["input"]), nl), int(c["offset"]), int(c["generator"]), c["order"])
            assert rc == 0 and orc.from_limbs(out) == I(c["output"])
        elif kind == "fft_multiply":
            rc, out = orc.fft_multiply_ref(f, orc.to_limbs(I(c["a"]), nl), orc.to_limbs(I(c["b"]), nl), int(c["omega"]))
            assert rc == 0 and orc.from_limbs(out) == I(c["output"])
        elif kind == "fast_multiply":
            rc, out = orc.fast_multiply_ref(f, orc.to_limbs(I(c["a"]), nl), orc.to_limbs(I(c["b"]), nl), int(c["root"]), c["root_order"])
            assert rc == 0 and orc.from_limbs(out) == I(c["output"])
    assert seen == {"ntt", "intt", "coset", "fft_multiply", "fast_multiply"}


def test_curve_rs_f631_kat():
    # curve.rs:494-495
    assert orc.ec_mul(1, (36, 60), 3, nl=1) == (617, 5)
    assert orc.ec_mul(1, (121, 387), 4, nl=1) == (121, 244)


def test_bn128_test_g1_relations():
    # bn128.rs:285-301
    G = (1, 2)
    assert orc.lib().orc_g1_on_curve(orc.ptr(orc.pts_to_arr([G]))) == 1
    G2 = orc.ec_mul(0, G, 2)
    lhs = orc.ec_add(0, orc.ec_add(0, G2, G), G)
    assert lhs == orc.ec_mul(0, G2, 2)
    assert orc.ec_add(0, orc.ec_mul(0, G, 9), orc.ec_mul(0, G, 5)) == orc.ec_add(0, orc.ec_mul(0, G, 12), orc.ec_mul(0, G, 2))
    assert orc.ec_mul(0, G, P_FR) == (0, 0)
    assert G2 == (1368015179489954701390400359078579693043519447331113978918064868415326638035,
                  9918110051302171585080402603319702774565515993150576347155970296011118125764)


def test_golden_curve_vectors():
    g = orc.golden("curve_vectors.json")
    for c in g["g1_mul"]:
        assert list(orc.ec_mul(0, (1, 2), int(c["k"]))) == I(c["out"])
    for c in g["g1_add"]:
        assert list(orc.ec_add(0, tuple(I(c["P"])), tuple(I(c["Q"])))) == I(c["out"])
    for c in g["msm"]:
        s = orc.to_limbs(I(c["scalars"]), 5)[:, :4] if c["tag"] == "unsanitized_scalar" else orc.to_limbs(I(c["scalars"]), 4)
        p = orc.pts_to_arr([tuple(x) for x in I(c["points"])])
        s = np.ascontiguousarray(s).reshape(-1, 4)
        p = p.reshape(-1, 8)
        assert list(orc.msm_ref(s, p)) == I(c["out"]), c["tag"]
        assert list(orc.msm_fast(s, p)) == I(c["out"]), c["tag"]


def test_golden_kzg():
    g = orc.golden("curve_vectors.json")
    for c in g["kzg"]:
        coef = orc.to_limbs(I(c["coef"]), 4)
        srs = orc.kzg_setup_ref(int(c["alpha"]), len(c["coef"]) - 1)
        assert [list(p) for p in orc.arr_to_pts(srs)] == I(c["srs"])
        assert list(orc.msm_ref(coef, srs)) == I(c["commit"])
        y, w = orc.kzg_open_ref(coef, int(c["u"]), srs)
        assert y == int(c["y"]) and list(w) == I(c["w"])
    # trapdoor identity: commit == [f(alpha)] G
    c = g["kzg"][1]
    fa = orc.poly_eval(FR, orc.to_limbs(I(c["coef"]), 4), int(c["alpha"]))
    assert list(orc.ec_mul(0, (1, 2), fa)) == I(c["commit"])
