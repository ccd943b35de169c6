"""CPU: the oracle's literal restatement of ntt::fast_zerofier / fast_evaluate / fast_interpolate (ntt.rs:118-252,
oracle/mzk_oracle.c) against the independent pure-Python transcription (tests/golden/poly_tree_vectors.json, made by
tests/golden/make_golden_poly.py) and against the mathematics (Z vanishes on the domain, evaluate = Horner,
interpolate o evaluate = id)."""
import numpy as np
import pytest
import orc
from orc import FR, M128

FID = {"fr": FR, "m128": M128}
CASES = orc.golden("poly_tree_vectors.json")


def _arr(fid, xs):
    return orc.to_limbs([int(x) for x in xs], orc.LIMBS[fid]) if len(xs) else np.zeros((0, orc.LIMBS[fid]), dtype=np.uint64)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%d-%d" % (c["field"], c["n"], c["root_order"]))
def test_oracle_matches_python_transcription(case):
    fid = FID[case["field"]]
    dom, vals, poly = _arr(fid, case["domain"]), _arr(fid, case["values"]), _arr(fid, case["poly"])
    root, order = int(case["root"]), case["root_order"]
    rc, z = orc.fast_zerofier_ref(fid, dom, root, order)
    assert rc == 0 and orc.from_limbs(z) == [int(x) for x in case["zerofier"]] if len(z) else case["zerofier"] == []
    rc, ev = orc.fast_evaluate_ref(fid, poly, dom, root, order)
    assert rc == 0 and (orc.from_limbs(ev) if len(ev) else []) == [int(x) for x in case["evaluate"]]
    rc, ip = orc.fast_interpolate_ref(fid, dom, vals, root, order)
    assert rc == 0 and (orc.from_limbs(ip) if len(ip) else []) == [int(x) for x in case["interpolate"]]


@pytest.mark.parametrize("fid", [FR, M128])
def test_oracle_properties_at_a_larger_size(fid):
    n, lg = 700, 11
    p = orc.MOD[fid]
    root = orc.root_of(fid, lg)
    dom = orc.synth_vector(fid, 4100, n)
    vals = orc.synth_vector(fid, 4101, n)
    rc, z = orc.fast_zerofier_ref(fid, dom, root, 1 << lg)
    assert rc == 0 and z.shape[0] == 1024                            # fast_multiply's untrimmed order
    zl = orc.from_limbs(z)
    assert zl[n] == 1 and not any(zl[n + 1:])
    for i in (0, 1, n // 2, n - 1):
        assert orc.poly_eval(fid, z, orc.from_limbs(dom[i:i + 1])[0]) == 0
    rc, ip = orc.fast_interpolate_ref(fid, dom, vals, root, 1 << lg)
    assert rc == 0 and ip.shape[0] <= n
    rc, back = orc.fast_evaluate_ref(fid, ip, dom, root, 1 << lg)
    assert rc == 0 and np.array_equal(back, vals)


def test_oracle_assertions():
    dom = orc.synth_vector(M128, 1, 4)
    assert orc.fast_zerofier_ref(M128, dom, orc.root_of(M128, 4), 8)[0] == -3      # root^order != 1
    assert orc.fast_zerofier_ref(M128, dom, orc.root_of(M128, 2), 8)[0] == -4      # not primitive
