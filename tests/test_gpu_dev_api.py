"""Device-resident entry points (*_dev) and the multi-GPU data path exercised on ONE GPU: every shard's
XYZZ partial is produced by the same calls bench.py makes per rank, the records are stacked the way
all_gather_into_tensor stacks them, and mzk_g1_fold_partials_dev folds them."""
import ctypes
import numpy as np
import pytest
import orc
from orc import FR, M128

pytestmark = pytest.mark.gpu


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


def _dp(t):
    return ctypes.c_void_p(t.data_ptr())


def _to_dev(torch, dev, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).reshape(-1).copy()).to(dev)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_partials_fold_equals_full_msm(env, world):
    torch, mz, L, dev, st = env
    from myzkp_amd import sharded
    n = 5000
    s = orc.synth_vector(FR, 60 + world, n)
    p = orc.synth_points(61 + world, n)
    p[17] = 0
    want = orc.msm_fast(s, p)
    recs = torch.zeros((world, 16), dtype=torch.int64, device=dev)
    recs_srs = torch.zeros((world, 16), dtype=torch.int64, device=dev)
    for r in range(world):
        lo, hi = sharded.shard_range(n, r, world)
        ds, dpnt = _to_dev(torch, dev, s[lo:hi]), _to_dev(torch, dev, p[lo:hi])
        _ok(L, L.mzk_msm_g1_bn254_partial_dev(_dp(ds), _dp(dpnt), ctypes.c_size_t(hi - lo), ctypes.c_void_p(recs[r].data_ptr()), st))
        h = ctypes.c_void_p()
        _ok(L, L.mzk_srs_from_device(_dp(dpnt), ctypes.c_size_t(hi - lo), ctypes.byref(h), st))
        _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(hi - lo), ctypes.c_void_p(recs_srs[r].data_ptr()), 1, st))
        torch.cuda.synchronize()
        L.mzk_srs_free(h)
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    for rr in (recs, recs_srs):
        _ok(L, L.mzk_g1_fold_partials_dev(_dp(rr), ctypes.c_int(world), _dp(out), st))
        torch.cuda.synchronize()
        assert mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == want


def test_partial_of_empty_and_cancelling_shards(env):
    torch, mz, L, dev, st = env
    p = orc.synth_points(5, 4)
    neg = p.copy()
    neg[:, 4:] = orc.to_limbs([orc.P_FQ - y for y in orc.from_limbs(p[:, 4:])], 4)
    s = orc.to_limbs([7, 8, 9, 10], 4)
    recs = torch.zeros((3, 16), dtype=torch.int64, device=dev)
    ds, dp1, dp2 = _to_dev(torch, dev, s), _to_dev(torch, dev, p), _to_dev(torch, dev, neg)
    _ok(L, L.mzk_msm_g1_bn254_partial_dev(_dp(ds), _dp(dp1), ctypes.c_size_t(4), ctypes.c_void_p(recs[0].data_ptr()), st))
    _ok(L, L.mzk_msm_g1_bn254_partial_dev(_dp(ds), _dp(dp2), ctypes.c_size_t(4), ctypes.c_void_p(recs[1].data_ptr()), st))
    _ok(L, L.mzk_msm_g1_bn254_partial_dev(_dp(ds), _dp(dp1), ctypes.c_size_t(0), ctypes.c_void_p(recs[2].data_ptr()), st))   # empty shard
    out = torch.ones(8, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_g1_fold_partials_dev(_dp(recs), ctypes.c_int(3), _dp(out), st))
    torch.cuda.synchronize()
    assert mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == (0, 0)     # P + (-P) + inf = inf


def test_ntt_dev_in_place_and_lde_and_fold_dev(env):
    torch, mz, L, dev, st = env
    lg = 13
    n = 1 << lg
    for fid, nl in ((FR, 4), (M128, 2)):
        v = orc.synth_vector(fid, 80, n)
        w = orc.root_of(fid, lg)
        root = mz.to_limbs([w], nl)
        d = _to_dev(torch, dev, v)
        _ok(L, L.mzk_ntt_dev(fid, root.ctypes.data_as(ctypes.c_void_p), _dp(d), _dp(d), ctypes.c_size_t(n), 0, st))   # in place
        torch.cuda.synchronize()
        rc, want = orc.ntt_fast(fid, w, v)
        assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(-1, nl), want)
        _ok(L, L.mzk_ntt_dev(fid, root.ctypes.data_as(ctypes.c_void_p), _dp(d), _dp(d), ctypes.c_size_t(n), 1, st))
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(-1, nl), v)
    # coset LDE and FRI fold, device-resident, M128
    coef = orc.synth_vector(M128, 81, n // 4)
    g = orc.m128_root(lg)
    d_c = _to_dev(torch, dev, coef)
    d_o = torch.zeros(n * 2, dtype=torch.int64, device=dev)
    off, gen = mz.to_limbs([orc.M128_GEN], 2), mz.to_limbs([g], 2)
    _ok(L, L.mzk_coset_lde_dev(M128, _dp(d_c), ctypes.c_size_t(n // 4), off.ctypes.data_as(ctypes.c_void_p), gen.ctypes.data_as(ctypes.c_void_p), _dp(d_o), ctypes.c_size_t(n), st))
    torch.cuda.synchronize()
    lde = d_o.cpu().numpy().view(np.uint64).reshape(-1, 2)
    assert np.array_equal(lde, mz.coset_lde(M128, coef, orc.M128_GEN, g, n))
    alpha = orc.from_limbs(orc.synth_vector(M128, 82, 1))[0]
    al = mz.to_limbs([alpha], 2)
    d_f = torch.zeros(n, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_fri_fold_dev(M128, _dp(d_o), ctypes.c_size_t(n), al.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), gen.ctypes.data_as(ctypes.c_void_p), _dp(d_f), st))
    torch.cuda.synchronize()
    assert np.array_equal(d_f.cpu().numpy().view(np.uint64).reshape(-1, 2), orc.fri_fold_ref(M128, lde, alpha, orc.M128_GEN, g))


@pytest.mark.parametrize("first,count", [(0, 1), (0, 37), (2047, 3), ((1 << 22) - 5, 40), ((1 << 33) - 7, 21), ((1 << 40) + 3, 17)])
def test_setup_range_any_offset_matches_fixed_base_oracle(env, first, count):
    """powers[i] = alpha^(first+i) * g1 (kzg.rs:33-36) for shard offsets on both sides of the alpha-table limit
    and counts that leave a ragged batch-inversion tail."""
    torch, mz, L, dev, st = env
    alpha = 0x1234567890abcdef1234567890abcdef % orc.P_FR
    g = (1, 2)
    scal = orc.to_limbs([pow(alpha, first + i, orc.P_FR) for i in range(count)], 4)
    want = orc.fixed_base_batch(g, scal)
    a_l, g_l = orc.to_limbs([alpha], 4), orc.pts_to_arr([g])
    out = torch.zeros((count, 8), dtype=torch.int64, device=dev)
    _ok(L, L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_size_t(first), ctypes.c_size_t(count), _dp(out), st))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


def test_setup_alpha_zero_gives_g_then_infinity(env):
    """alpha = 0: powers = [g, inf, inf, ...] -- the batched inversion must pass infinities through."""
    torch, mz, L, dev, st = env
    a_l, g_l = orc.to_limbs([0], 4), orc.pts_to_arr([(1, 2)])
    out = torch.ones((20, 8), dtype=torch.int64, device=dev)
    _ok(L, L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_size_t(0), ctypes.c_size_t(20), _dp(out), st))
    torch.cuda.synchronize()
    o = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(o[0], g_l[0]) and not o[1:].any()


def test_setup_wide_table_path_full_array_vs_oracle(env):
    """count >= 2^16 switches to the 16-bit fixed-base table (k_fb_table16 + k_fb_powers<16>): every power of a
    ragged, offset range over a non-standard base must equal the oracle's fixed-base ladder."""
    torch, mz, L, dev, st = env
    alpha = 0xfeedfacecafebeef0123456789abcdef0fedcba987654321 % orc.P_FR
    g = orc.ec_mul(0, (1, 2), 5)
    first, count = 12345, (1 << 16) + 37
    acc, scal = pow(alpha, first, orc.P_FR), []
    for _ in range(count):
        scal.append(acc); acc = acc * alpha % orc.P_FR
    want = orc.fixed_base_batch(g, orc.to_limbs(scal, 4))
    a_l, g_l = orc.to_limbs([alpha], 4), orc.pts_to_arr([g])
    out = torch.zeros((count, 8), dtype=torch.int64, device=dev)
    _ok(L, L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_size_t(first), ctypes.c_size_t(count), _dp(out), st))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("c", [12, 16, 20, 22])
def test_srs_tables_with_explicit_window_width(env, c):
    """the merged-bucket path at the window widths used for large SRS sizes (20 bits from 2^21+1 points, 22 from
    2^23+1), forced here on a small SRS: 254 / c + 1 tables, 2^(c-1) buckets, two-level sort with up to 8192 buckets per bin"""
    torch, mz, L, dev, st = env
    n = 1 << 13
    p = orc.synth_points(90 + c, n)
    p[5] = 0
    s = orc.synth_vector(FR, 91 + c, n)
    s[7] = orc.to_limbs([orc.P_FR - 1], 4)[0]
    s[8] = 0
    s[9] = orc.to_limbs([(1 << (c - 1))], 4)[0]          # exactly half the window: the signed-digit boundary
    s[10] = orc.to_limbs([(1 << c) - 1], 4)[0]
    want = orc.msm_fast(s, p)
    dp, ds = _to_dev(torch, dev, p), _to_dev(torch, dev, s)
    h = ctypes.c_void_p()
    _ok(L, L.mzk_srs_from_device_ex(_dp(dp), ctypes.c_size_t(n), ctypes.c_int(c), ctypes.byref(h), st))
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    for m in (n, n - 3, 100):
        _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(m), _dp(out), 0, st))
        torch.cuda.synchronize()
        assert mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == (want if m == n else orc.msm_fast(s[:m], p[:m])), (c, m)
    L.mzk_srs_free(h)


def test_table_budget_degrades_the_layout_not_the_result(env):
    """VERDICT r03 missing #5: commit_kzg never fails for lack of memory, so a handle whose window tables do not fit
    (mzk_set_table_budget, or a refused hipMalloc) degrades -- every 2nd table with two bucket sets, every 4th with four, the
    prepared points alone -- and every commitment, opening and saved/loaded handle stays bit-identical to the oracle's."""
    torch, mz, L, dev, st = env
    L.mzk_srs_table_bytes.restype = ctypes.c_size_t
    n = (1 << 16) + 37
    p = orc.synth_points(601, n)
    p[11] = 0
    s = orc.synth_vector(FR, 602, n)
    s[3] = 0
    want = orc.msm_fast(s, p)
    want_short = orc.msm_fast(s[:5000], p[:5000])
    dp, ds = _to_dev(torch, dev, p), _to_dev(torch, dev, s)
    full = 16 * n * 64                                           # 16-bit windows below 2^19 points: 16 tables
    seen = []
    try:
        for budget, exp_sets in ((0, 1), (full, 1), (full - 1, 2), (8 * n * 64 - 1, 4), (4 * n * 64 - 1, 0), (1, 0)):
            _ok(L, L.mzk_set_table_budget(ctypes.c_size_t(budget)))
            h = ctypes.c_void_p()
            _ok(L, L.mzk_srs_from_device(_dp(dp), ctypes.c_size_t(n), ctypes.byref(h), st))
            sets = L.mzk_srs_bucket_sets(h)
            assert sets == exp_sets, (budget, sets)
            assert L.mzk_srs_window_bits(h) == (16 if exp_sets else 0)
            rows = {1: 16, 2: 8, 4: 4, 0: 2}[exp_sets]
            assert L.mzk_srs_table_bytes(h) == rows * n * 64
            if budget > 1:
                assert L.mzk_srs_table_bytes(h) <= budget
            out = torch.zeros(8, dtype=torch.int64, device=dev)
            _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(n), _dp(out), 0, st))
            torch.cuda.synchronize()
            assert mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == want, budget
            _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(5000), _dp(out), 0, st))
            torch.cuda.synchronize()
            assert mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == want_short, budget
            # the XYZZ partial of a shard (multi-GPU path) folds to the same point
            part = torch.zeros(16, dtype=torch.int64, device=dev)
            _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(n), _dp(part), 1, st))
            _ok(L, L.mzk_g1_fold_partials_dev(_dp(part), 1, _dp(out), st))
            torch.cuda.synchronize()
            assert mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == want, budget
            seen.append(sets)
            L.mzk_srs_free(h)
        # explicit widths degrade the same way; 20-bit windows fall back to 17 bits with two sets
        _ok(L, L.mzk_set_table_budget(ctypes.c_size_t(10 * n * 64)))
        h = ctypes.c_void_p()
        _ok(L, L.mzk_srs_from_device_ex(_dp(dp), ctypes.c_size_t(n), ctypes.c_int(20), ctypes.byref(h), st))
        assert (L.mzk_srs_window_bits(h), L.mzk_srs_bucket_sets(h)) == (17, 2)
        out = torch.zeros(8, dtype=torch.int64, device=dev)
        _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(ds), ctypes.c_size_t(n), _dp(out), 0, st))
        torch.cuda.synchronize()
        assert mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == want
        L.mzk_srs_free(h)
    finally:
        L.mzk_set_table_budget(ctypes.c_size_t(0))
    assert seen == [1, 1, 2, 4, 0, 0]


def test_workspace_gives_itself_back(env):
    """the workspace only ever grew.  A 2^24-pair generic MSM leaves gigabytes in its slots; a workspace budget below
    that releases them at once, a full-table 2^20 handle built afterwards gets its undegraded layout and commits bit-exactly
    (oracle Pippenger on the same pairs); with the budget in force a later call of another kind trims what the MSM left instead of
    stacking on top of it; mzk_trim_workspace gives everything back and the next calls rebuild what they need."""
    torch, mz, L, dev, st = env
    L.mzk_srs_table_bytes.restype = ctypes.c_size_t
    big, n = 1 << 24, 1 << 20
    try:
        pts = torch.empty(big * 8, dtype=torch.int64, device=dev)
        sc = torch.empty(big * 4, dtype=torch.int64, device=dev)
        _ok(L, L.mzk_synth_g1_points_dev(ctypes.c_uint64(811), ctypes.c_size_t(big), _dp(pts), st))
        _ok(L, L.mzk_synth_field_dev(FR, ctypes.c_uint64(812), ctypes.c_size_t(big), _dp(sc), st))
        out = torch.zeros(8, dtype=torch.int64, device=dev)
        _ok(L, L.mzk_msm_g1_bn254_dev(_dp(sc), _dp(pts), ctypes.c_size_t(big), _dp(out), st))
        torch.cuda.synchronize()
        held = mz.workspace_bytes()
        assert held > (1 << 30), held                              # prepared points + endomorphism images alone are 4 GiB
        budget = 256 << 20
        mz.set_workspace_budget(budget)
        assert mz.workspace_bytes() <= budget
        h = ctypes.c_void_p()
        _ok(L, L.mzk_srs_from_device(_dp(pts), ctypes.c_size_t(n), ctypes.byref(h), st))
        assert L.mzk_srs_bucket_sets(h) == 1 and L.mzk_srs_window_bits(h) == 17 and L.mzk_srs_table_bytes(h) == 15 * n * 64
        _ok(L, L.mzk_kzg_commit_srs_dev(h, _dp(sc), ctypes.c_size_t(n), _dp(out), 0, st))
        torch.cuda.synchronize()
        got = mz.array_to_points(out.cpu().numpy().view(np.uint64))[0]
        hp = pts[: n * 8].cpu().numpy().view(np.uint64).reshape(n, 8)
        hs = sc[: n * 4].cpu().numpy().view(np.uint64).reshape(n, 4)
        assert got == orc.msm_fast(hs, hp)
        L.mzk_srs_free(h)
        # under the budget a call of another kind releases what the commit left before it grows its own slots
        after_commit = mz.workspace_bytes()
        v = torch.empty((1 << 22) * 4, dtype=torch.int64, device=dev)
        w = torch.empty_like(v)
        _ok(L, L.mzk_synth_field_dev(FR, ctypes.c_uint64(813), ctypes.c_size_t(1 << 22), _dp(v), st))
        root = mz.to_limbs([mz.root_of_unity(FR, 22)], 4)
        _ok(L, L.mzk_ntt_dev(FR, root.ctypes.data_as(ctypes.c_void_p), _dp(v), _dp(w), ctypes.c_size_t(1 << 22), 0, st))
        torch.cuda.synchronize()
        assert mz.workspace_bytes() <= max(budget, (1 << 22) * 32 + 4096), (after_commit, mz.workspace_bytes())
        back = torch.empty_like(v)
        released = mz.trim_workspace()
        assert released > 0 and mz.workspace_bytes() == 0
        _ok(L, L.mzk_ntt_dev(FR, root.ctypes.data_as(ctypes.c_void_p), _dp(w), _dp(back), ctypes.c_size_t(1 << 22), 1, st))      # plans rebuilt
        torch.cuda.synchronize()
        assert torch.equal(back, v)
    finally:
        mz.set_workspace_budget(0)


def test_interpolation_plans_and_pool_count_as_workspace(env):
    """the polynomial routines' scratch pool and their cached interpolation plans (up to four per context) are workspace
    like every other buffer: mzk_workspace_bytes sees them, mzk_trim_workspace counts them in what it gives back, and the next
    interpolation over the same domain rebuilds its plan and returns the same polynomial (ntt.rs:185-252)."""
    torch, mz, L, dev, st = env
    fid = orc.M128
    n = 1 << 10
    p = orc.MOD[fid]
    w = orc.root_of(fid, 10)
    dom = mz.to_limbs([pow(w, i, p) for i in range(n - 5)], 2)
    vals = orc.synth_vector(fid, 4711, n - 5)
    mz.trim_workspace()
    base = mz.workspace_bytes()
    first = mz.fast_interpolate(fid, dom, vals, w, n)
    held = mz.workspace_bytes()
    assert held > base, (base, held)                  # the plan (subproduct tree levels, 1 / Z'(d_i)) and the parked scratch blocks
    released = mz.trim_workspace()
    assert released >= held - base and mz.workspace_bytes() == 0
    again = mz.fast_interpolate(fid, dom, vals, w, n)
    assert np.array_equal(np.asarray(first), np.asarray(again))
    rc, want = orc.fast_interpolate_ref(fid, dom, vals, w, n)
    assert rc == 0 and np.array_equal(np.asarray(again), want)


def test_many_commit_wide_tables_respect_the_table_budget(env):
    """the 12-bit tables the grid-batched pass builds once per handle (from 2^13 coefficients on) fall under
    mzk_set_table_budget like the handle's own tables; a refusal is remembered on the handle, leaves no error message behind on a call
    that succeeds, and the commitments are the same points (das/eigenda.rs:92-101 -> kzg.rs:57-59)."""
    torch, mz, L, dev, st = env
    L.mzk_srs_table_bytes.restype = ctypes.c_size_t
    L.mzk_last_error.restype = ctypes.c_char_p
    nn, count = 1 << 13, 4
    pt = torch.empty(nn * 8, dtype=torch.int64, device=dev)
    cf = torch.empty(count * nn * 4, dtype=torch.int64, device=dev)
    _ok(L, L.mzk_synth_g1_points_dev(ctypes.c_uint64(931), ctypes.c_size_t(nn), _dp(pt), st))
    _ok(L, L.mzk_synth_field_dev(FR, ctypes.c_uint64(932), ctypes.c_size_t(count * nn), _dp(cf), st))
    outs = []
    try:
        for budget in (0, 1):            # 1 byte: nothing beyond what the handle needs to work at all
            L.mzk_set_table_budget(ctypes.c_size_t(0))
            h = ctypes.c_void_p()
            _ok(L, L.mzk_srs_from_device(_dp(pt), ctypes.c_size_t(nn), ctypes.byref(h), st))
            own = L.mzk_srs_table_bytes(h)
            L.mzk_set_table_budget(ctypes.c_size_t(own if budget else 0))       # the handle's own tables fit exactly, nothing more does
            o = torch.zeros(count * 8, dtype=torch.int64, device=dev)
            for _ in range(2):           # the second call must not retry the refused allocation (same result either way)
                _ok(L, L.mzk_kzg_commit_srs_many_dev(h, _dp(cf), ctypes.c_size_t(nn), ctypes.c_size_t(count), _dp(o), st))
            torch.cuda.synchronize()
            grown = L.mzk_srs_table_bytes(h) - own
            assert (grown == 0) if budget else (grown > 0), (budget, own, grown)
            if budget:
                assert L.mzk_last_error() in (b"", None) or b"wide window tables" not in L.mzk_last_error()
            outs.append(o.cpu().numpy().copy())
            L.mzk_srs_free(h)
    finally:
        L.mzk_set_table_budget(ctypes.c_size_t(0))
    assert np.array_equal(outs[0], outs[1])
    hp = pt.cpu().numpy().view(np.uint64).reshape(nn, 8)
    hc = cf.cpu().numpy().view(np.uint64).reshape(count, nn, 4)
    got = mz.array_to_points(outs[0].view(np.uint64).reshape(count, 8))
    assert got[0] == orc.msm_fast(hc[0], hp) and got[-1] == orc.msm_fast(hc[-1], hp)


def test_second_host_thread_gets_busy_not_corruption():
    """include/mzk.h: one host thread at a time.  A call arriving while another thread is inside the library returns MZK_E_BUSY
    (-11) before touching any state; the call in progress is unaffected."""
    import threading
    import myzkp_amd as mzz
    mzz.init(0)
    Lb = mzz.lib()
    lg = 22
    n = 1 << lg
    v = orc.synth_vector(orc.FR, 31337, n)
    w = orc.root_of(orc.FR, lg)
    wl = mzz.to_limbs([w], 4)
    out = np.zeros_like(v)
    small_in = orc.synth_vector(orc.FR, 5, 64)
    small_w = mzz.to_limbs([orc.root_of(orc.FR, 6)], 4)
    rc_rec, ok_small, stop = [], [], threading.Event()
    rc, want_small = orc.ntt_fast(orc.FR, orc.root_of(orc.FR, 6), small_in)

    def hammer():
        o = np.zeros_like(small_in)
        while not stop.is_set():
            r = Lb.mzk_ntt(0, small_w.ctypes.data_as(ctypes.c_void_p), small_in.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(64), 0)
            rc_rec.append(r)
            if r == 0:
                ok_small.append(bool(np.array_equal(o, want_small)))
    th = threading.Thread(target=hammer)
    th.start()
    try:
        done, tries = 0, 0
        while done < 6 and tries < 100000:      # ~10 ms each with the transfers: long enough for the other thread to run into it
            r = Lb.mzk_ntt(0, wl.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), 0)
            assert r in (0, -11)                 # -11: the other thread was inside at that moment -- repeat, as the header says
            done += r == 0
            tries += 1
        assert done == 6
    finally:
        stop.set()
        th.join()
    assert set(rc_rec) <= {0, -11} and -11 in rc_rec, "the second thread never met a call in progress"
    assert all(ok_small)
    # the big transform, alone again, is right
    assert Lb.mzk_ntt(0, wl.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), 0) == 0
    rc, want = orc.ntt_fast(orc.FR, w, v)
    assert rc == 0 and np.array_equal(out, want)


def test_init_and_ctx_select_from_a_second_thread_cannot_flip_the_context_under_a_batch_call():
    """mzk_init's idempotent fast path used to run outside the entry guard and re-select context 0 -- a second
    thread calling it (wrappers init per thread) could switch the current context under the CtxScope switches of a batch call
    in progress.  Now it gets MZK_E_BUSY like every other entry point; the batch's points stay right."""
    import threading
    import myzkp_amd as mzz
    mzz.init_devices([0, 0, 0])
    Lb = mzz.lib()
    n, count = 1 << 15, 6                       # 2^15 coefficients: wide tables -> one commit per lane (three contexts)
    p = orc.synth_points(4711, n)
    h = mzz.Srs(p)
    coefs = np.stack([orc.synth_vector(orc.FR, 900 + k, n) for k in range(count)])
    want = [orc.msm_fast(coefs[k], p) for k in range(count)]
    rcs, stop = [], threading.Event()

    def hammer():
        while not stop.is_set():
            rcs.append(Lb.mzk_init(0))
            rcs.append(Lb.mzk_ctx_select(2))
    th = threading.Thread(target=hammer)
    th.start()
    try:
        done, tries = 0, 0
        out = np.zeros((count, 8), dtype=np.uint64)
        while done < 8 and tries < 100000:
            r = Lb.mzk_kzg_commit_srs_batch(h._h, coefs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), ctypes.c_size_t(count), out.ctypes.data_as(ctypes.c_void_p))
            assert r in (0, -11), Lb.mzk_last_error()
            if r == 0:
                assert mzz.array_to_points(out) == want
                done += 1
            tries += 1
        assert done == 8
    finally:
        stop.set()
        th.join()
    assert set(rcs) <= {0, -11} and -11 in rcs, "the second thread never met a call in progress"
    h.close()
    mzz.init_devices([0])
