"""Row-cooperative group operations (myzkp_amd/csrc/mzk_row.h: one XYZZ addition / doubling per wave, field elements spread over
DPP rows) against the plain exception-complete formulas of mzk_ec.h, which the host build pins on the oracle's affine group law
(curve.rs:44-161; tests/test_hostcheck_arith.py).  Pairs cover independent points, P + P, P + (-P) and infinity on either side;
the MSM parity tests then exercise the same code inside the bucket-reduction tails, the window Horner and the partial fold.
The integer model of the row product's column / carry bounds is tests/test_row_product_model.py (CPU)."""
import ctypes
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,reps", [(1, 1 << 14, 1), (0xabcdef, 4096, 17), (77, 8, 3), (5, 1, 0)])
def test_row_add_and_dbl_equal_the_plain_formulas(seed, n, reps):
    import myzkp_amd as mz
    mz.init(0)
    L = mz.lib()
    bad = ctypes.c_uint64(123)
    rc = L.mzk_selftest_row_ec(ctypes.c_uint64(seed), ctypes.c_size_t(n), ctypes.c_int(reps), ctypes.byref(bad))
    assert rc == 0, L.mzk_last_error()
    assert bad.value == 0


@pytest.mark.parametrize("seed,n", [(1, 1 << 14), (99, 8), (5, 1)])
def test_wave_inversion_equals_the_single_lane_safegcd(seed, n):
    import myzkp_amd as mz
    mz.init(0)
    L = mz.lib()
    bad = ctypes.c_uint64(123)
    rc = L.mzk_selftest_inv_wave(ctypes.c_uint64(seed), ctypes.c_size_t(n), ctypes.byref(bad))
    assert rc == 0, L.mzk_last_error()
    assert bad.value == 0


def test_row_and_quad_tails_give_the_same_commitments():
    """MZK_ROW_TAILS=0 keeps the DPP-quad tails selectable; both must give the oracle's point on every path that has a tail:
    small three-launch commits, the general pipeline with tables, the generic layout's window Horner, partial records."""
    import os, subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import orc, myzkp_amd as mz
mz.init(0)
out = []
for n in (3, 300, 5000, 1 << 15):
    s, p = orc.synth_vector(orc.FR, 900 + n, n), orc.synth_points(901 + n, n)
    h = mz.Srs(p)
    out.append((n, mz.msm_g1(s, p), h.commit(s)))
    h.close()
print(repr(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, MZK_ROW_TAILS=flag)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = eval(r.stdout.strip().splitlines()[-1])
    assert res["0"] == res["1"]
    import orc
    for n, generic, commit in res["1"]:
        s, p = orc.synth_vector(orc.FR, 900 + n, n), orc.synth_points(901 + n, n)
        assert generic == commit == orc.msm_fast(s, p), n
