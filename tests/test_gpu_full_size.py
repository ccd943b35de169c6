"""BASELINE.json's full-size configurations under `-m gpu`.

  configs[3]  2^24-pair G1 MSM: generic layout (GLV, no tables) AND an SRS handle with its 16 window tables (16 GiB),
              both against the closed form  sum_i s_i [alpha^i]G = [f(alpha)]G  (polynomial.rs:156-165, kzg.rs:57-59)
  configs[1]  2^24-point NTT over Fr and M128: the whole output vector against the oracle's iterative transform,
              four outputs against direct evaluation (ntt.rs:7-48: out[k] = f(w^k)), and intt(ntt(x)) == x
  configs[4]  end-to-end KZG at degree 2^22 with all three trapdoor identities (kzg.rs:27-72)
  skew        2^22 pairs with bit-vector / repeated scalars: 256-entry segments (lgseg = 8) and the heavy-bucket
              combine run together
At these sizes the MSM takes code paths no smaller test reaches: 8-byte sort records, lgseg up to 10, point references
up to 2^28, S / per_fine of the two-level sort."""
import ctypes
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import pytest
import orc
from orc import FR, M128, P_FR

pytestmark = pytest.mark.gpu
G = (1, 2)


@pytest.fixture(scope="module")
def env():
    import torch
    import myzkp_amd as mz
    mz.init(0)
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return torch, mz, L, dev, st


def _ok(L, rc):
    assert rc == 0, L.mzk_last_error().decode()


def _dp(t, off=0):
    return ctypes.c_void_p(t.data_ptr() + off)


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _point(mz, t):
    return mz.array_to_points(t.cpu().numpy().view(np.uint64))[0]


def _srs_points_dev(env, alpha, n):
    """[alpha^i] G, i < n, built on the device (setup_kzg, kzg.rs:27-40)."""
    torch, mz, L, dev, st = env
    pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
    a_l, g_l = mz.to_limbs([alpha], 4), mz.points_to_array([G])
    _ok(L, L.mzk_kzg_setup_g1_dev(_vp(a_l), _vp(g_l), ctypes.c_size_t(n - 1), _dp(pts), st))
    return pts


def _to_dev(torch, dev, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).reshape(-1)).to(dev)


def test_msm_2_24_generic_and_srs_tables_trapdoor_identity(env):
    torch, mz, L, dev, st = env
    lg = 24
    n = 1 << lg
    alpha = orc.from_limbs(orc.synth_vector(FR, 2401, 1))[0]
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(2402), ctypes.c_size_t(n), _dp(sc), st))
    pts = _srs_points_dev(env, alpha, n)
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    # generic layout: arbitrary points, GLV split, 8 bucket sets, window Horner
    _ok(L, L.mzk_msm_g1_bn254_dev(_dp(sc), _dp(pts), ctypes.c_size_t(n), _dp(out), st))
    # fixed-base layout: 16 window tables of 2^24 points each
    h = ctypes.c_void_p()
    _ok(L, L.mzk_srs_from_device(_dp(pts), ctypes.c_size_t(n), ctypes.byref(h), st))
    del pts
    _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(sc), ctypes.c_size_t(n), _dp(out, 64), ctypes.c_int(0), st))
    torch.cuda.synchronize()
    s_cpu = orc.synth_vector(FR, 2402, n)
    assert np.array_equal(s_cpu.view(np.int64).reshape(-1)[:4096], sc[:4096].cpu().numpy())
    want = orc.ec_mul(0, G, orc.poly_eval(FR, s_cpu, alpha))
    assert _point(mz, out[:8]) == want, "generic 2^24 MSM != [f(alpha)]G"
    assert _point(mz, out[8:]) == want, "SRS-table 2^24 commit != [f(alpha)]G"
    # a ragged prefix through the same tables (n not a multiple of any tile / segment size)
    m = n - 12345
    _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(sc), ctypes.c_size_t(m), _dp(out), ctypes.c_int(0), st))
    torch.cuda.synchronize()
    assert _point(mz, out[:8]) == orc.ec_mul(0, G, orc.poly_eval(FR, s_cpu[:m], alpha))
    L.mzk_srs_free(h)
    torch.cuda.empty_cache()


def test_msm_generic_first_size_of_the_19_bit_windows_trapdoor_identity(env):
    """3 x 2^21 pairs is where the generic (GLV) layout goes from 16-bit to 19-bit windows (seven windows per half-scalar over 7 x 2^18
    buckets, sorted as 2^21 with an empty eighth window: mzk_msm.hip choose_shape_glv): a ragged size just above the switch and the
    last one below it, against the closed form sum_i s_i [alpha^i]G = [f(alpha)]G (polynomial.rs:156-165)."""
    torch, mz, L, dev, st = env
    n = (3 << 21) + 4321
    assert L.mzk_msm_generic_window_bits(ctypes.c_size_t(n)) == 19 and L.mzk_msm_generic_window_bits(ctypes.c_size_t((3 << 21) - 1)) == 16
    alpha = orc.from_limbs(orc.synth_vector(FR, 2301, 1))[0]
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(2302), ctypes.c_size_t(n), _dp(sc), st))
    pts = _srs_points_dev(env, alpha, n)
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    s_cpu = orc.synth_vector(FR, 2302, n)
    for k, m in enumerate((n, (3 << 21) - 1)):
        _ok(L, L.mzk_msm_g1_bn254_dev(_dp(sc), _dp(pts), ctypes.c_size_t(m), _dp(out, 64 * k), st))
        torch.cuda.synchronize()
        assert _point(mz, out[8 * k:8 * k + 8]) == orc.ec_mul(0, G, orc.poly_eval(FR, s_cpu[:m], alpha)), "generic MSM of %d pairs != [f(alpha)]G" % m
    del sc, pts
    torch.cuda.empty_cache()


@pytest.mark.parametrize("lg", [20, 24])
def test_msm_unstructured_points_generic_and_tables_vs_oracle_pippenger(env, lg):
    """SURVEY 8d configs 3 / 4 on UNSTRUCTURED points: 2^20 and 2^24 try-and-increment hash-to-curve
    points (mzk_synth_g1_points_dev, the bench's generator; no trapdoor, no closed form), uniform scalars, through the generic
    GLV layout and through window tables (default width for the size, and BASELINE's 16 bits at 2^20), against the oracle's
    Pippenger on the same streams regenerated on the CPU."""
    torch, mz, L, dev, st = env
    n = 1 << lg
    threads = orc.usable_threads()
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(5200 + lg), ctypes.c_size_t(n), _dp(sc), st))
    _ok(L, L.mzk_synth_g1_points_dev(ctypes.c_uint64(5300 + lg), ctypes.c_size_t(n), _dp(pts), st))
    out = torch.zeros(8 * 3, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_msm_g1_bn254_dev(_dp(sc), _dp(pts), ctypes.c_size_t(n), _dp(out), st))
    widths = [1] + ([16] if lg == 20 else [])
    for k, wb in enumerate(widths):
        h = ctypes.c_void_p()
        _ok(L, L.mzk_srs_from_device_ex(_dp(pts), ctypes.c_size_t(n), ctypes.c_int(wb), ctypes.byref(h), st))
        _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(sc), ctypes.c_size_t(n), _dp(out, 64 * (k + 1)), ctypes.c_int(0), st))
        torch.cuda.synchronize()
        L.mzk_srs_free(h)
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    # the oracle works on the SAME scalars and points: scalars regenerated on the CPU, points downloaded (1 GiB in 20 ms; the
    # try-and-increment generator costs a modular square root per point on the CPU -- half a minute at 2^24); the generators
    # themselves are the oracle's streams bit for bit: whole vectors at 2^20 in bench.py, a 2^14 suffix here
    s_cpu = orc.synth_vector(FR, 5200 + lg, n, threads)
    p_cpu = pts.cpu().numpy().view(np.uint64).reshape(-1, 8)
    assert np.array_equal(s_cpu.view(np.int64).reshape(-1), sc.cpu().numpy())
    tail = 1 << 14
    p_tail = orc.synth_points(5300 + lg, n, threads)[-tail:] if lg <= 20 else None
    if p_tail is not None:
        assert np.array_equal(p_tail, p_cpu[-tail:])
    want = orc.msm_fast(s_cpu, p_cpu, threads)
    assert _point(mz, out[:8]) == want, "generic MSM on unstructured points != oracle Pippenger"
    for k, wb in enumerate(widths):
        assert _point(mz, out[8 * (k + 1):8 * (k + 2)]) == want, "window-table commit (option %d) on unstructured points != oracle Pippenger" % wb
    del sc, pts
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fid,nl", [(FR, 4), (M128, 2)])
def test_ntt_2_24_full_vector_vs_oracle_and_direct_evaluation(env, fid, nl):
    torch, mz, L, dev, st = env
    lg = 24
    n = 1 << lg
    w = orc.root_of(fid, lg)
    root = mz.to_limbs([w], nl)
    x = torch.empty(n * nl, dtype=torch.int64, device=dev)
    y = torch.empty(n * nl, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_synth_field_dev(fid, ctypes.c_uint64(2410 + fid), ctypes.c_size_t(n), _dp(x), st))
    _ok(L, L.mzk_ntt_dev(fid, _vp(root), _dp(x), _dp(y), ctypes.c_size_t(n), 0, st))
    torch.cuda.synchronize()
    got = y.cpu().numpy().view(np.uint64).reshape(-1, nl)
    v = orc.synth_vector(fid, 2410 + fid, n)
    rc, want = orc.ntt_fast(fid, w, v)
    assert rc == 0 and np.array_equal(got, want), "2^24 NTT differs from the oracle's transform"
    # out[k] = f(w^k) (ntt.rs:40-46) at four positions by Horner over all 2^24 coefficients -- a different algorithm
    ks = [1, 0x5a5a5a, n // 2 + 3, n - 1]
    with ThreadPoolExecutor(4) as ex:
        direct = list(ex.map(lambda k: orc.poly_eval(fid, v, pow(w, k, orc.MOD[fid])), ks))
    assert [orc.from_limbs(got[k:k + 1])[0] for k in ks] == direct
    # inverse in place: intt(ntt(x)) == x
    _ok(L, L.mzk_ntt_dev(fid, _vp(root), _dp(y), _dp(y), ctypes.c_size_t(n), 1, st))
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    del x, y
    torch.cuda.empty_cache()


def test_e2e_kzg_degree_2_22_all_three_identities(env):
    """BASELINE configs[4] at full size: evaluations -> iNTT -> setup -> commit -> open, device-resident."""
    torch, mz, L, dev, st = env
    lg = 22
    n = 1 << lg
    ev = torch.empty(n * 4, dtype=torch.int64, device=dev)
    cf = torch.empty(n * 4, dtype=torch.int64, device=dev)
    out = torch.zeros(20, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(2422), ctypes.c_size_t(n), _dp(ev), st))
    w = orc.root_of(FR, lg)
    root = mz.to_limbs([w], 4)
    _ok(L, L.mzk_ntt_dev(mz.FIELD_FR, _vp(root), _dp(ev), _dp(cf), ctypes.c_size_t(n), 1, st))
    alpha = orc.from_limbs(orc.synth_vector(FR, 2423, 1))[0]
    u = orc.from_limbs(orc.synth_vector(FR, 2424, 1))[0]
    u_l = mz.to_limbs([u], 4)
    pts = _srs_points_dev(env, alpha, n)
    for with_tables in (0, 1):      # the one-shot pipeline (plain prepared points) and the table handle
        h = ctypes.c_void_p()
        _ok(L, L.mzk_srs_from_device_ex(_dp(pts), ctypes.c_size_t(n), ctypes.c_int(with_tables), ctypes.byref(h), st))
        _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(cf), ctypes.c_size_t(n), _dp(out), ctypes.c_int(0), st))
        _ok(L, L.mzk_kzg_open_srs_dev(h, _dp(cf), ctypes.c_size_t(n), _vp(u_l), _dp(out, 64), _dp(out, 96), st))
        torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint64)
        commit, y, wpt = mz.array_to_points(o[:8])[0], mz.from_limbs(o[8:12].reshape(1, 4))[0], mz.array_to_points(o[12:20])[0]
        if with_tables == 0:
            rc, coef_cpu = orc.ntt_fast(FR, w, orc.synth_vector(FR, 2422, n), inverse=True)
            assert rc == 0 and np.array_equal(coef_cpu.view(np.int64).reshape(-1), cf.cpu().numpy())
            fa, fu = orc.poly_eval(FR, coef_cpu, alpha), orc.poly_eval(FR, coef_cpu, u)
            want_c = orc.ec_mul(0, G, fa)
            want_w = orc.ec_mul(0, G, (fa - fu) * pow(alpha - u, -1, P_FR) % P_FR)
        assert commit == want_c, with_tables          # commit == [f(alpha)] G
        assert y == fu, with_tables                   # y == f(u)
        assert wpt == want_w, with_tables             # w == [(f(alpha) - y) / (alpha - u)] G
        L.mzk_srs_free(h)
    del pts
    torch.cuda.empty_cache()


def test_msm_2_22_skewed_scalars_long_segments_and_heavy_buckets(env):
    """Witness-like scalars at 2^22 pairs: a bit vector, one value repeated, small values, and uniform ones.  The merged
    layout runs with 256-entry segments (lgseg = 8); buckets 0 and the repeated digits collect hundreds of thousands of
    entries and go through the heavy-bucket combine; the generic layout sees the same skew after the GLV split."""
    torch, mz, L, dev, st = env
    lg = 22
    n = 1 << lg
    alpha = orc.from_limbs(orc.synth_vector(FR, 2431, 1))[0]
    s = orc.synth_vector(FR, 2432, n)
    q = n // 4
    s[:q] = 0
    s[:q, 0] = (orc.synth_vector(M128, 2433, q)[:, 0] & np.uint64(1))            # bit vector
    s[q:2 * q] = orc.to_limbs([0x1234567890abcdef1234567890abcdef0fedcba987654321], 4)[0]    # one repeated value
    s[2 * q:3 * q, 1:] = 0
    s[2 * q:3 * q, 0] &= np.uint64(0xffff)                                         # 16-bit values
    pts = _srs_points_dev(env, alpha, n)
    ds = _to_dev(torch, dev, s)
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_msm_g1_bn254_dev(_dp(ds), _dp(pts), ctypes.c_size_t(n), _dp(out), st))
    h = ctypes.c_void_p()
    _ok(L, L.mzk_srs_from_device(_dp(pts), ctypes.c_size_t(n), ctypes.byref(h), st))
    _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(n), _dp(out, 64), ctypes.c_int(0), st))
    torch.cuda.synchronize()
    want = orc.ec_mul(0, G, orc.poly_eval(FR, s, alpha))
    assert _point(mz, out[:8]) == want
    assert _point(mz, out[8:]) == want
    L.mzk_srs_free(h)
    del pts
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pattern", ["ones", "equal", "200_values", "600_values", "ones_then_uniform"])
def test_msm_2_20_every_form_of_the_heavy_bucket_combine(env, pattern):
    """k_seg_combine_heavy has two forms: few deferred buckets (at most half its grid: bit vectors, constant polynomials) are shared
    by several workgroups each, with a last-arrival sum over their scratch records; many of them get one workgroup each, in turn.
    1 bucket (all ones: every workgroup on it), 15 (all scalars equal), 200 values (two or three workgroups per bucket), 600
    (one per bucket), and ones followed by uniform scalars; both layouts, trapdoor identity."""
    torch, mz, L, dev, st = env
    n = 1 << 20
    alpha = orc.from_limbs(orc.synth_vector(FR, 5511, 1))[0]
    s = np.zeros((n, 4), dtype=np.uint64)
    if pattern == "ones":
        s[:, 0] = 1
    elif pattern == "equal":
        s[:] = orc.to_limbs([0x2abcdef01234567890abcdef1234567890abcdef0fedcba9876543210fedcba9 % P_FR], 4)[0]
    elif pattern.endswith("_values"):
        k = int(pattern.split("_")[0])
        s[:, 0] = (orc.synth_vector(M128, 5512, n)[:, 0] % np.uint64(k)) * np.uint64(97) + np.uint64(1)     # k distinct values below 2^16
    else:
        s[: n // 2, 0] = 1
        s[n // 2:] = orc.synth_vector(FR, 5513, n // 2)
    pts = _srs_points_dev(env, alpha, n)
    ds = _to_dev(torch, dev, s)
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_msm_g1_bn254_dev(_dp(ds), _dp(pts), ctypes.c_size_t(n), _dp(out), st))
    h = ctypes.c_void_p()
    _ok(L, L.mzk_srs_from_device(_dp(pts), ctypes.c_size_t(n), ctypes.byref(h), st))
    _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(n), _dp(out, 64), ctypes.c_int(0), st))
    _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(n), _dp(out, 64), ctypes.c_int(0), st))      # twice: the arrival counters start from zero again
    torch.cuda.synchronize()
    want = orc.ec_mul(0, G, orc.poly_eval(FR, s, alpha))
    assert _point(mz, out[:8]) == want
    assert _point(mz, out[8:]) == want
    L.mzk_srs_free(h)
    del pts
    torch.cuda.empty_cache()
