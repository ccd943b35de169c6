"""The largest sizes the ABI accepts (include/mzk.h "Size limits"), beyond BASELINE's 2^24: a 2^27-pair MSM (generic layout and
16 window tables of 2^27 points: 2^31 table entries) against the closed form  sum_i s_i [alpha^i]G = [f(alpha)]G
(polynomial.rs:156-165, kzg.rs:57-59), and 2^28-point transforms over Fr (its whole 2-adicity) and M128 against the closed
form of the all-ones polynomial on a coset,  sum_j a^j w^(jk) = (a^n - 1) / (a w^k - 1)  (ntt.rs:254-269), the plain transform
of a random vector against the LDE with offset 1, and intt(ntt(x)) == x.  No CPU transform of that size is needed."""
import ctypes
import numpy as np
import pytest
import orc
from orc import FR, M128

pytestmark = pytest.mark.gpu
G = (1, 2)


@pytest.fixture(scope="module")
def env():
    import torch
    import myzkp_amd as mz
    mz.init(0)
    return torch, mz, mz.lib(), torch.device("cuda", 0), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ok(L, rc):
    assert rc == 0, L.mzk_last_error().decode()


def _dp(t, off=0):
    return ctypes.c_void_p(t.data_ptr() + off)


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _elem(t, k, nl):
    return orc.from_limbs(t[k * nl:(k + 1) * nl].cpu().numpy().view(np.uint64).reshape(1, nl))[0]


@pytest.mark.parametrize("fid,nl,a", [(FR, 4, 0x1234567), (M128, 2, 3)])
def test_ntt_2_28_closed_form_and_round_trip(env, fid, nl, a):
    torch, mz, L, dev, st = env
    lg = 28
    n, p = 1 << lg, orc.MOD[fid]
    w = orc.root_of(fid, lg)
    root, one_l, off = mz.to_limbs([w], nl), mz.to_limbs([1], nl), mz.to_limbs([a], nl)
    x = torch.zeros(n * nl, dtype=torch.int64, device=dev)
    y = torch.empty(n * nl, dtype=torch.int64, device=dev)
    x.view(-1, nl)[:, 0] = 1
    _ok(L, L.mzk_coset_lde_dev(fid, _dp(x), ctypes.c_size_t(n), _vp(off), _vp(root), _dp(y), ctypes.c_size_t(n), st))
    num = (pow(a, n, p) - 1) % p
    rng = np.random.default_rng(2800 + fid)
    for k in [0, 1, n - 1, n // 2, n // 2 + 1] + [int(k) for k in rng.integers(0, n, 43)]:
        assert _elem(y, k, nl) == num * pow((a * pow(w, k, p) - 1) % p, -1, p) % p, "k=%d" % k
    _ok(L, L.mzk_synth_field_dev(fid, ctypes.c_uint64(2828), ctypes.c_size_t(n), _dp(x), st))
    _ok(L, L.mzk_ntt_dev(fid, _vp(root), _dp(x), _dp(y), ctypes.c_size_t(n), 0, st))
    z = torch.empty(n * nl, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_coset_lde_dev(fid, _dp(x), ctypes.c_size_t(n), _vp(one_l), _vp(root), _dp(z), ctypes.c_size_t(n), st))
    torch.cuda.synchronize()
    assert torch.equal(y, z)
    del z
    _ok(L, L.mzk_ntt_dev(fid, _vp(root), _dp(y), _dp(y), ctypes.c_size_t(n), 1, st))
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    del x, y
    torch.cuda.empty_cache()


def test_msm_2_27_generic_and_window_tables_trapdoor_identity(env):
    torch, mz, L, dev, st = env
    lg = 27
    n = 1 << lg
    alpha = orc.from_limbs(orc.synth_vector(FR, 2701, 1))[0]
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(2702), ctypes.c_size_t(n), _dp(sc), st))
    pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_kzg_setup_g1_dev(_vp(mz.to_limbs([alpha], 4)), _vp(mz.points_to_array([G])), ctypes.c_size_t(n - 1), _dp(pts), st))
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_msm_g1_bn254_dev(_dp(sc), _dp(pts), ctypes.c_size_t(n), _dp(out), st))
    h = ctypes.c_void_p()
    _ok(L, L.mzk_srs_from_device(_dp(pts), ctypes.c_size_t(n), ctypes.byref(h), st))
    _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(sc), ctypes.c_size_t(n), _dp(out, 64), ctypes.c_int(0), st))
    # one pair more than the limit is refused before anything is read
    assert L.mzk_msm_g1_bn254_dev(_dp(sc), _dp(pts), ctypes.c_size_t(n + 1), _dp(out), st) == -1   # MZK_E_ARG
    torch.cuda.synchronize()
    s_cpu = orc.synth_vector(FR, 2702, n)
    assert np.array_equal(s_cpu.view(np.int64).reshape(-1)[-4096:], sc[-4096:].cpu().numpy())
    want = orc.ec_mul(0, G, orc.poly_eval(FR, s_cpu, alpha))
    assert mz.array_to_points(out[:8].cpu().numpy().view(np.uint64))[0] == want, "generic 2^27 MSM != [f(alpha)]G"
    assert mz.array_to_points(out[8:].cpu().numpy().view(np.uint64))[0] == want, "2^27 commit against window tables != [f(alpha)]G"
    L.mzk_srs_free(h)
    del pts, sc
    torch.cuda.empty_cache()
