"""Oracle restatements of the "next" rows (FRI split-and-fold fri.rs:182-193, batch_open_kzg
kzg.rs:74-88, prove_degree_bound kzg.rs:121-134) against the Python golden vectors.  CPU only."""
import numpy as np
import orc
from orc import FR, M128, I


def test_fri_fold_golden():
    for c in orc.golden("fri_vectors.json")["cases"]:
        f = {"Fr": FR, "M128": M128}[c["field"]]
        out = orc.fri_fold_ref(f, orc.to_limbs(I(c["input"]), orc.LIMBS[f]), int(c["alpha"]), int(c["offset"]), int(c["omega"]))
        assert orc.from_limbs(out) == I(c["output"])


def test_batch_open_and_degree_bound_golden():
    g = orc.golden("curve_vectors.json")
    for c in g["kzg_batch_open"]:
        srs = orc.pts_to_arr([tuple(x) for x in I(c["srs"])])
        ys, w = orc.kzg_batch_open_ref(orc.to_limbs(I(c["coef"]), 4), I(c["us"]), srs)
        assert ys == I(c["ys"]) and list(w) == I(c["w"])
    for c in g["kzg_degree_bound"]:
        srs = orc.pts_to_arr([tuple(x) for x in I(c["srs"])])
        rc, out = orc.kzg_degree_bound_ref(orc.to_limbs(I(c["coef"]), 4), srs, c["d"])
        assert rc == 0 and list(out) == I(c["out"])
    # degree above the bound does not fit the SRS (index panic), d > max_d underflows
    srs = orc.pts_to_arr([tuple(x) for x in I(g["kzg_degree_bound"][0]["srs"])])
    coef = orc.to_limbs(I(g["kzg_degree_bound"][0]["coef"]), 4)
    assert orc.kzg_degree_bound_ref(coef, srs, 3)[0] == -5
    assert orc.kzg_degree_bound_ref(coef, srs, 17)[0] == -5
