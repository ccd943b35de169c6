"""GPU parity: G1 MSM / KZG setup, commit, open through the C ABI vs the oracle and golden vectors.
Bit-exact on the canonical affine output."""
import numpy as np
import pytest
import orc
from orc import FR, FQ, P_FR, P_FQ, I

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


def test_golden_msm_cases(mz):
    g = orc.golden("curve_vectors.json")
    for c in g["msm"]:
        nl = 5 if c["tag"] == "unsanitized_scalar" else 4
        s = np.ascontiguousarray(orc.to_limbs(I(c["scalars"]), nl)[:, :4]).reshape(-1, 4)
        p = orc.pts_to_arr([tuple(x) for x in I(c["points"])]).reshape(-1, 8)
        assert list(mz.msm_g1(s, p)) == I(c["out"]), c["tag"]


def test_golden_kzg(mz):
    g = orc.golden("curve_vectors.json")
    for c in g["kzg"]:
        coef = orc.to_limbs(I(c["coef"]), 4)
        srs = mz.kzg_setup_g1(int(c["alpha"]), len(c["coef"]) - 1)
        assert [list(p) for p in orc.arr_to_pts(srs)] == I(c["srs"])
        assert list(mz.kzg_commit(coef, srs)) == I(c["commit"])
        y, w = mz.kzg_open(coef, int(c["u"]), srs)
        assert y == int(c["y"]) and list(w) == I(c["w"])
        h = mz.Srs(srs)
        assert list(h.commit(coef)) == I(c["commit"])
        h.close()


@pytest.mark.parametrize("n", [1, 2, 3, 17, 255, 256, 257, 1000, 1024])
def test_msm_vs_literal_oracle(mz, n):
    # config[0]: degree-2^10 polynomial on the reference-algorithm CPU path
    s = orc.synth_vector(FR, 1000 + n, n)
    p = orc.synth_points(2000 + n, n)
    want = orc.msm_ref(s, p) if n <= 300 else orc.msm_fast(s, p)
    assert mz.msm_g1(s, p) == want


@pytest.mark.parametrize("lg", [12, 14, 16, 18])
def test_msm_vs_fast_oracle(mz, lg):
    n = 1 << lg
    s = orc.synth_vector(FR, 31 + lg, n)
    p = orc.synth_points(77 + lg, n)
    assert mz.msm_g1(s, p) == orc.msm_fast(s, p, threads=8)


def test_edge_batch(mz):
    # SURVEY 8d-3 edge batch: 0, 1, r-1, repeated scalars, all-equal points, P and -P adjacent
    n = 4096
    s = orc.synth_vector(FR, 5, n)
    p = orc.synth_points(6, n)
    sl = orc.from_limbs(s)
    sl[0], sl[1], sl[2] = 0, 1, P_FR - 1
    for i in range(10, 200):
        sl[i] = sl[10]                      # repeated scalars
    s = orc.to_limbs(sl, 4)
    p[300:600] = p[300]                     # all-equal points
    for i in range(300, 600):
        s[i] = s[300]                       # ... with equal scalars: bucket sees P + P
    p[700] = p[701]
    p[700, 4:] = orc.to_limbs([P_FQ - orc.from_limbs(p[701:702, 4:])[0]], 4)[0]   # -P next to P
    s[700] = s[701]
    p[900:910] = 0                          # points at infinity
    want = orc.msm_fast(s, p)
    assert mz.msm_g1(s, p) == want


def test_all_same_bucket_skew(mz):
    # adversarial skew: every scalar equal -> one bucket per window holds all n points
    n = 2048
    p = orc.synth_points(9, n)
    s = orc.to_limbs([0x1234567890abcdef1234567890abcdef] * n, 4)
    assert mz.msm_g1(s, p) == orc.msm_fast(s, p)


def test_trapdoor_identity_2pow16(mz):
    # commit == [f(alpha)] G on an SRS built on the GPU; w == [q(alpha)] G
    n = 1 << 16
    alpha = orc.from_limbs(orc.synth_vector(FR, 404, 1))[0]
    srs = mz.kzg_setup_g1(alpha, n - 1)
    f = orc.synth_vector(FR, 405, n)
    fa = orc.poly_eval(FR, f, alpha)
    assert mz.kzg_commit(f, srs) == orc.ec_mul(0, (1, 2), fa)
    # spot-check the SRS itself against the oracle's fixed-base ladder
    for i in (0, 1, 2, 255, 256, 4097, n - 1):
        assert orc.arr_to_pts(srs[i:i + 1])[0] == orc.ec_mul(0, (1, 2), pow(alpha, i, P_FR))
    u = orc.from_limbs(orc.synth_vector(FR, 406, 1))[0]
    y, w = mz.kzg_open(f, u, srs)
    assert y == orc.poly_eval(FR, f, u)
    # q(alpha) = (f(alpha) - y) / (alpha - u)
    qa = (fa - y) * pow(alpha - u, -1, P_FR) % P_FR
    assert w == orc.ec_mul(0, (1, 2), qa)


def test_msm_linearity_2pow20(mz):
    # BASELINE size, size-independent property: MSM(s, P) + MSM(t, P) == MSM(s + t, P), and the
    # structured-SRS closed form
    n = 1 << 20
    alpha = 0x1234567
    srs = mz.kzg_setup_g1(alpha, n - 1)
    s = orc.synth_vector(FR, 11, n)
    fa = orc.poly_eval(FR, s, alpha)
    assert mz.msm_g1(s, srs) == orc.ec_mul(0, (1, 2), fa)


def test_srs_window_tables_path(mz):
    # device-resident SRS with precomputed window tables T[w][i] = 2^(16 w) P_i (n >= 2^14): all windows
    # share one bucket set; result must equal the plain MSM
    n = 1 << 14
    p = orc.synth_points(321, n)
    p[100:104] = 0                                  # infinity entries in the SRS
    h = mz.Srs(p)
    for seed, m in ((1, n), (2, n - 5), (3, 1000), (4, 1)):
        s = orc.synth_vector(FR, 500 + seed, m)
        assert h.commit(s) == orc.msm_fast(s, p[:m])
    # edge scalars through the merged-bucket path
    sl = [0, 1, P_FR - 1, (1 << 15), (1 << 15) + 1, (1 << 16) - 1, P_FR - (1 << 15)] + [0x7fff8000_7fff8000] * 9
    s = orc.to_limbs(sl, 4)
    assert h.commit(s) == orc.msm_fast(s, p[:len(sl)])
    with pytest.raises(mz.MzkError) as e:
        h.commit(orc.synth_vector(FR, 9, n + 1))
    assert e.value.code == -5
    h.close()


def _skewed_scalar_sets(n, seed):
    """realistic skew: bit vectors and repeated values put most entries into a handful of buckets"""
    rng = np.random.default_rng(seed)
    uni = orc.synth_vector(FR, seed, n)
    ones = np.zeros((n, 4), dtype=np.uint64); ones[:, 0] = 1
    bits = np.zeros((n, 4), dtype=np.uint64); bits[:, 0] = rng.integers(0, 2, n, dtype=np.uint64)
    equal = np.tile(uni[:1], (n, 1))
    small = np.zeros((n, 4), dtype=np.uint64); small[:, 0] = uni[:, 0] & np.uint64(0xffff)
    mixed = uni.copy(); mixed[: (3 * n) // 4] = ones[: (3 * n) // 4]        # 75 % ones, 25 % uniform
    minus1 = np.tile(orc.to_limbs([P_FR - 1], 4), (n, 1))                     # every digit negative-carrying
    # short scalars (witness values): every entry of a window lands in one or a few coarse bins of the two-level sort (fine_plan)
    byte = np.zeros((n, 4), dtype=np.uint64); byte[:, 0] = uni[:, 0] & np.uint64(0xff)
    b128 = uni.copy(); b128[:, 2:] = 0
    b248 = uni.copy(); b248[:, 3] &= np.uint64((1 << 56) - 1)
    return {"ones": ones, "bits": bits, "equal": equal, "small16": small, "mixed": mixed, "minus_one": minus1, "bytes": byte, "128_bit": b128, "248_bit": b248}


def test_skewed_scalars_merged_two_level_sort(mz):
    """SRS (merged-bucket) path at a size that takes the two-level sort, the aggregated LDS counters and the
    heavy-bucket combine: result must equal the oracle for every skew pattern."""
    n = 1 << 14
    p = orc.synth_points(777, n)
    h = mz.Srs(p)
    for name, s in _skewed_scalar_sets(n, 31).items():
        assert h.commit(s) == orc.msm_fast(s, p), name
    h.close()


@pytest.mark.parametrize("n", [5, 700, 1024, 3000, 4096, 8191, 16384])
def test_small_commit_without_sort_launch_skewed_scalars(mz, n):
    """Commits of at most 2^14 coefficients against narrow window tables (8 / 10 bits) find every bucket's entries by
    walking the scalars inside the accumulate kernel (k_small_accumulate_scan, no sort launch): uniform scalars, and
    the skew patterns that overflow the per-bucket LDS list (all scalars equal: one bucket per window gets n entries,
    'ones': one bucket gets all of them), infinity entries in the SRS included."""
    p = orc.synth_points(4242 + n, n)
    if n > 100:
        p[7:9] = 0
    h = mz.Srs(p)
    sets = _skewed_scalar_sets(n, 77 + n)
    sets["uniform"] = orc.synth_vector(FR, 78 + n, n)
    sets["same_digit_everywhere"] = np.tile(orc.to_limbs([sum(5 << (8 * w) for w in range(31))], 4), (n, 1))
    for name, s in sets.items():
        assert h.commit(s) == orc.msm_fast(s, p), name
    h.close()


def test_skewed_scalars_generic_two_level_sort(mz):
    """generic layout takes the two-level sort from n = 2^19 (c = 16); heavy buckets there go through
    k_seg_combine_wide's deferral; short scalars leave its coarse bins (window-major keys) very unevenly filled (fine_plan)"""
    n = 1 << 19
    p = orc.synth_points(778, n)
    sets = _skewed_scalar_sets(n, 32)
    for name in ("ones", "bits", "mixed", "equal", "bytes", "small16", "128_bit", "248_bit"):
        assert mz.msm_g1(sets[name], p) == orc.msm_fast(sets[name], p), name


def test_glv_edge_scalars_generic_path(mz):
    """the generic MSM splits every scalar as k = k1 + k2 lambda (mzk_glv.h) and runs on (P, phi(P)): scalars at the
    decomposition's corners (lambda itself, its negative, powers of two, values that make one half zero or negative)"""
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    ks = [0, 1, 2, lam, lam - 1, lam + 1, P_FR - lam, (lam * lam) % P_FR, P_FR - 1, P_FR - 2, 1 << 126, 1 << 127, (1 << 128) - 1,
          1 << 253, (P_FR - 1) // 2, (P_FR + 1) // 2, 9931322734385697763, 147946756881789319010696353538189108491,
          147946756881789319000765030803803410728]
    pts = orc.synth_points(4242, len(ks))
    s = orc.to_limbs(ks, 4)
    assert mz.msm_g1(s, pts) == orc.msm_fast(s, pts)
    for i, k in enumerate(ks):                      # one at a time: each is a bare scalar multiplication
        assert mz.msm_g1(s[i:i + 1], pts[i:i + 1]) == orc.ec_mul(0, orc.arr_to_pts(pts[i:i + 1])[0], k), hex(k)


@pytest.mark.parametrize("lg", [13, 15, 16, 17])
def test_generic_msm_window_shapes(mz, lg):
    """sizes that select different GLV window shapes (c = 12 / 13 / 13 / 16) and both sort paths"""
    n = 1 << lg
    p = orc.synth_points(600 + lg, n)
    s = orc.synth_vector(FR, 601 + lg, n)
    assert mz.msm_g1(s, p) == orc.msm_fast(s, p)


def test_randomised_differential_generic_and_srs(mz):
    """25 seeded random cases: ragged sizes across every window shape, mixed scalar patterns (uniform, short, bit,
    repeated, r-1, zero), infinity points sprinkled in, same result through the generic MSM and an SRS handle"""
    import random
    rnd = random.Random(20261001)
    base = orc.synth_points(99, 6000)
    for case in range(25):
        n = rnd.choice([1, 2, 3, 7, 31, 100, 255, 256, 257, 1000, 2049, 4095, 4096, 4097, 5999])
        p = base[:n].copy()
        vals = []
        for i in range(n):
            kind = rnd.randrange(8)
            vals.append([rnd.randrange(P_FR), rnd.getrandbits(16), rnd.getrandbits(1), 7, P_FR - 1, 0, rnd.getrandbits(128),
                         P_FR - rnd.getrandbits(20) - 1][kind])
        for i in rnd.sample(range(n), min(n, 3)):
            if rnd.random() < 0.5:
                p[i] = 0
        s = orc.to_limbs(vals, 4)
        want = orc.msm_fast(s, p)
        assert mz.msm_g1(s, p) == want, (case, n)
        h = mz.Srs(p)
        assert h.commit(s) == want, (case, n)
        h.close()


def test_wide_sort_records_above_2pow20(mz):
    """just above 2^20 pairs the two-level sort switches from 4-byte to 8-byte intermediate records (point reference + fine
    key + sign no longer fit 32 bits): trapdoor identity through the generic MSM and through an SRS handle with tables"""
    n = (1 << 20) + 4096
    alpha = 0xabcdef123
    srs = mz.kzg_setup_g1(alpha, n - 1)
    s = orc.synth_vector(FR, 2020, n)
    want = orc.ec_mul(0, (1, 2), orc.poly_eval(FR, s, alpha))
    assert mz.msm_g1(s, srs) == want
    h = mz.Srs(srs)
    assert h.commit(s) == want
    h.close()


@pytest.mark.parametrize("n", [5, (1 << 18), (1 << 20) - 12345])
def test_host_buffer_calls_in_pieces_equal_the_resident_calls(mz, n):
    """commit_kzg(&poly, &pk) hands over host Vecs (kzg.rs:57-59), so the host-buffer entry points upload their
    inputs in pieces and sort / accumulate a piece while the next one crosses PCIe (msm_chunked_impl: per-piece bucket arrays, summed
    and reduced once).  From 2^18 pairs on; below that one piece as before.  A ragged size, the smallest chunked size and a tiny one
    through both entry points against the device-resident calls on the same pairs (bit-identical) and the oracle's Pippenger."""
    import ctypes
    import torch
    L, dev = mz.lib(), torch.device("cuda", 0)
    L.mzk_last_error.restype = ctypes.c_char_p
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    assert L.mzk_synth_field_dev(0, ctypes.c_uint64(7100 + n % 97), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st) == 0
    assert L.mzk_synth_g1_points_dev(ctypes.c_uint64(7200 + n % 97), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st) == 0
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    assert L.mzk_msm_g1_bn254_dev(ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), st) == 0
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
    assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr() + 64), 0, st) == 0
    torch.cuda.synchronize()
    hs = sc.cpu().numpy().view(np.uint64).reshape(n, 4).copy()
    hp = pt.cpu().numpy().view(np.uint64).reshape(n, 8).copy()
    want = orc.msm_fast(hs, hp)
    res = mz.array_to_points(out.cpu().numpy().view(np.uint64).reshape(2, 8))
    assert res[0] == want and res[1] == want
    assert mz.msm_g1(hs, hp) == want, "host-buffer MSM"
    got = np.zeros((1, 8), dtype=np.uint64)
    for _ in range(2):          # twice: the second call reuses every workspace slot of the first
        assert L.mzk_kzg_commit_srs(h, hs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), got.ctypes.data_as(ctypes.c_void_p)) == 0, L.mzk_last_error()
        assert mz.array_to_points(got)[0] == want, "host-scalar commit"
    # a prefix of the coefficients against the same handle (n smaller than the SRS: the pieces follow n, the table rows the handle)
    m = n - n // 3
    assert L.mzk_kzg_commit_srs(h, hs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(m), got.ctypes.data_as(ctypes.c_void_p)) == 0
    assert mz.array_to_points(got)[0] == orc.msm_fast(hs[:m], hp[:m])
    L.mzk_srs_free(h)
