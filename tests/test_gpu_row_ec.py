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


def test_tuning_build_switches_give_the_same_commitments_and_transforms():
    """The shipped library reads no environment variable (tests/test_abi_load.py).  The A/B switches live in the tuning build
    (python -m myzkp_amd.build --tuning, loaded through MZK_HIP_LIB): every non-default path they select -- the DPP-quad tails
    of round 2, the sorted small path, direct-store coarse scatter, the prefetching accumulate, un-fused NTT edges, Montgomery-only
    twiddles -- must still give the shipped library's results, which are the oracle's."""
    import os, subprocess, sys
    import myzkp_amd.build as b
    tuning = b.build(tuning=True)
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import orc, myzkp_amd as mz
mz.init(0)
out = []
for n in (3, 300, 5000, 1 << 15, 1 << 19):
    s, p = orc.synth_vector(orc.FR, 900 + n, n), orc.synth_points(901 + n, n)
    h = mz.Srs(p)
    out.append((n, mz.msm_g1(s, p), h.commit(s)))
    h.close()
for fid, lg in ((orc.FR, 13), (orc.FR, 20), (orc.M128, 20), (orc.FR, 21)):
    v = orc.synth_vector(fid, 77 + lg, 1 << lg)
    out.append((fid, lg, mz.ntt(fid, orc.root_of(fid, lg), v).tobytes().hex()[:4096], int(mz.ntt(fid, orc.root_of(fid, lg), v).sum(dtype=np.uint64))))
print(repr(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    variants = {"shipped": {},
                "quad_tails": {"MZK_ROW_TAILS": "0"},
                "sorted_small_direct_scatter_prefetch": {"MZK_SMALL_SCAN": "0", "MZK_COARSE_STAGED": "0", "MZK_ACC_PREFETCH": "1"},
                "ntt_unfused_montgomery": {"MZK_NTT_FUSE_EDGES": "0", "MZK_NTT_SHOUP": "0"}}
    for name, extra in variants.items():
        env = dict(os.environ, **extra)
        if name != "shipped":
            env["MZK_HIP_LIB"] = tuning
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = eval(r.stdout.strip().splitlines()[-1])
    for name in variants:
        assert res[name] == res["shipped"], name
    import orc
    for rec in res["shipped"]:
        if len(rec) == 3:
            n, generic, commit = rec
            s, p = orc.synth_vector(orc.FR, 900 + n, n), orc.synth_points(901 + n, n)
            assert generic == commit == orc.msm_fast(s, p), n
