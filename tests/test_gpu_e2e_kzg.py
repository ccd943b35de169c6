"""BASELINE configs[4]: end-to-end KZG on the device -- evaluations -> iNTT -> coefficients -> setup(alpha)
-> commit -> open(u), everything resident in HBM, verified with the trapdoor identities:
commit == [f(alpha)] G,  y == f(u),  w == [(f(alpha) - y) / (alpha - u)] G."""
import ctypes
import numpy as np
import pytest
import orc
from orc import FR, P_FR

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lg", [10, 16, 18])
def test_end_to_end_device_resident(lg):
    import torch
    import myzkp_amd as mz
    mz.init(0)
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 1 << lg

    def ok(rc):
        assert rc == 0, L.mzk_last_error().decode()

    def dp(t):
        return ctypes.c_void_p(t.data_ptr())

    evals = torch.empty(n * 4, dtype=torch.int64, device=dev)
    coef = torch.empty(n * 4, dtype=torch.int64, device=dev)
    srs_pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
    out = torch.zeros(8 + 4 + 8, dtype=torch.int64, device=dev)
    ok(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(77 + lg), ctypes.c_size_t(n), dp(evals), st))
    root = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lg)], 4)
    ok(L.mzk_ntt_dev(mz.FIELD_FR, root.ctypes.data_as(ctypes.c_void_p), dp(evals), dp(coef), ctypes.c_size_t(n), 1, st))
    alpha = orc.from_limbs(orc.synth_vector(FR, 900 + lg, 1))[0]
    u = orc.from_limbs(orc.synth_vector(FR, 901 + lg, 1))[0]
    a_l, u_l, g_l = mz.to_limbs([alpha], 4), mz.to_limbs([u], 4), mz.points_to_array([(1, 2)])
    ok(L.mzk_kzg_setup_g1_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n - 1), dp(srs_pts), st))
    h = ctypes.c_void_p()
    ok(L.mzk_srs_from_device(dp(srs_pts), ctypes.c_size_t(n), ctypes.byref(h), st))
    ok(L.mzk_kzg_commit_srs_dev(h, dp(coef), ctypes.c_size_t(n), dp(out), 0, st))
    ok(L.mzk_kzg_open_srs_dev(h, dp(coef), ctypes.c_size_t(n), u_l.ctypes.data_as(ctypes.c_void_p),
                              ctypes.c_void_p(out.data_ptr() + 64), ctypes.c_void_p(out.data_ptr() + 96), st))
    torch.cuda.synchronize()
    o = out.cpu().numpy().view(np.uint64)
    commit = mz.array_to_points(o[:8])[0]
    y = mz.from_limbs(o[8:12].reshape(1, 4))[0]
    w = mz.array_to_points(o[12:20])[0]
    # CPU side: coefficients via the oracle's iNTT, then the trapdoor identities
    ev_cpu = orc.synth_vector(FR, 77 + lg, n)
    rc, coef_cpu = orc.ntt_fast(FR, mz.from_limbs(root)[0], ev_cpu, inverse=True)
    assert rc == 0 and np.array_equal(coef_cpu.view(np.int64).reshape(-1), coef.cpu().numpy())
    fa = orc.poly_eval(FR, coef_cpu, alpha)
    assert commit == orc.ec_mul(0, (1, 2), fa)
    assert y == orc.poly_eval(FR, coef_cpu, u)
    qa = (fa - y) * pow(alpha - u, -1, P_FR) % P_FR
    assert w == orc.ec_mul(0, (1, 2), qa)
    # and the SRS itself at a few indices
    pts = srs_pts.cpu().numpy().view(np.uint64).reshape(-1, 8)
    for i in (0, 1, n // 2, n - 1):
        assert orc.arr_to_pts(pts[i:i + 1])[0] == orc.ec_mul(0, (1, 2), pow(alpha, i, P_FR))
    L.mzk_srs_free(h)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_end_to_end_on_one_gpu(world):
    """The N-GPU form of configs[4], every rank's calls replayed on one GPU: replicated iNTT and quotient,
    rank g builds SRS powers [lo, hi) and MSMs its slice of the coefficients / of the quotient; the XYZZ
    partials are folded as after an all-gather."""
    import torch
    import myzkp_amd as mz
    from myzkp_amd import sharded
    mz.init(0)
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lg = 15
    n = 1 << lg

    def ok(rc):
        assert rc == 0, L.mzk_last_error().decode()

    def dp(t, off_bytes=0):
        return ctypes.c_void_p(t.data_ptr() + off_bytes)

    evals = torch.empty(n * 4, dtype=torch.int64, device=dev)
    coef = torch.empty(n * 4, dtype=torch.int64, device=dev)
    quo = torch.zeros(n * 4, dtype=torch.int64, device=dev)
    yv = torch.zeros(4, dtype=torch.int64, device=dev)
    ok(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(4242), ctypes.c_size_t(n), dp(evals), st))
    root = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lg)], 4)
    ok(L.mzk_ntt_dev(mz.FIELD_FR, root.ctypes.data_as(ctypes.c_void_p), dp(evals), dp(coef), ctypes.c_size_t(n), 1, st))
    alpha = orc.from_limbs(orc.synth_vector(FR, 4243, 1))[0]
    u = orc.from_limbs(orc.synth_vector(FR, 4244, 1))[0]
    a_l, u_l, g_l = mz.to_limbs([alpha], 4), mz.to_limbs([u], 4), mz.points_to_array([(1, 2)])
    ok(L.mzk_kzg_open_quotient_dev(dp(coef), ctypes.c_size_t(n), u_l.ctypes.data_as(ctypes.c_void_p), dp(yv), dp(quo), st))
    rec_c = torch.zeros((world, 16), dtype=torch.int64, device=dev)
    rec_w = torch.zeros((world, 16), dtype=torch.int64, device=dev)
    for r in range(world):
        lo, hi = sharded.shard_range(n, r, world)
        pts = torch.empty((hi - lo) * 8, dtype=torch.int64, device=dev)
        ok(L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.c_size_t(lo), ctypes.c_size_t(hi - lo), dp(pts), st))
        h = ctypes.c_void_p()
        ok(L.mzk_srs_from_device(dp(pts), ctypes.c_size_t(hi - lo), ctypes.byref(h), st))
        ok(L.mzk_kzg_commit_srs_dev(h, dp(coef, lo * 32), ctypes.c_size_t(hi - lo), ctypes.c_void_p(rec_c[r].data_ptr()), 1, st))
        qhi = min(hi, n - 1)                      # the quotient has n - 1 coefficients
        ok(L.mzk_kzg_commit_srs_dev(h, dp(quo, lo * 32), ctypes.c_size_t(max(qhi - lo, 0)), ctypes.c_void_p(rec_w[r].data_ptr()), 1, st))
        torch.cuda.synchronize()
        L.mzk_srs_free(h)
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    ok(L.mzk_g1_fold_partials_dev(dp(rec_c), ctypes.c_int(world), dp(out), st))
    ok(L.mzk_g1_fold_partials_dev(dp(rec_w), ctypes.c_int(world), dp(out, 64), st))
    torch.cuda.synchronize()
    o = out.cpu().numpy().view(np.uint64)
    commit, w = mz.array_to_points(o[:8])[0], mz.array_to_points(o[8:16])[0]
    y = mz.from_limbs(yv.cpu().numpy().view(np.uint64).reshape(1, 4))[0]
    coef_cpu = coef.cpu().numpy().view(np.uint64).reshape(-1, 4)
    fa = orc.poly_eval(FR, coef_cpu, alpha)
    assert commit == orc.ec_mul(0, (1, 2), fa)
    assert y == orc.poly_eval(FR, coef_cpu, u)
    assert w == orc.ec_mul(0, (1, 2), (fa - y) * pow(alpha - u, -1, P_FR) % P_FR)


@pytest.mark.parametrize("n,world", [(1000, 3), (1 << 16, 8), (33, 2), (5, 4)])
def test_sharded_open_quotient_equals_the_whole_quotient(n, world):
    """the quotient of open_kzg (kzg.rs:61-72) computed slice by slice -- mzk_kzg_open_slice_value_dev, the top-down
    carries (sharded.open_carries), mzk_kzg_open_slice_quotient_dev -- with all `world` ranks inside one process: the concatenated
    slices are the quotient mzk_kzg_open_quotient_dev computes in one piece (followed by q[n-1] = b_n = 0), y is the same f(u), and
    both equal the recurrence b_i = c_i + u b_{i+1} on Python integers."""
    import torch
    import myzkp_amd as mz
    from myzkp_amd import sharded
    mz.init(0)
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    coef = torch.empty(n * 4, dtype=torch.int64, device=dev)
    assert L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(4100 + n % 89), ctypes.c_size_t(n), ctypes.c_void_p(coef.data_ptr()), st) == 0
    u = orc.from_limbs(orc.synth_vector(FR, 4200 + n % 89, 1))[0]
    u_l = mz.to_limbs([u], 4)
    whole_y = torch.zeros(4, dtype=torch.int64, device=dev)
    whole_q = torch.zeros(n * 4, dtype=torch.int64, device=dev)             # n - 1 elements written, the last stays 0 = b_n
    assert L.mzk_kzg_open_quotient_dev(ctypes.c_void_p(coef.data_ptr()), ctypes.c_size_t(n), u_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(whole_y.data_ptr()),
                                       ctypes.c_void_p(whole_q.data_ptr()), st) == 0, L.mzk_last_error().decode()
    ops = sharded.DeviceOpenOps()
    spans = [sharded.shard_range(n, g, world) for g in range(world)]
    values = []
    for lo, hi in spans:
        v = ops.slice_value(coef[lo * 4:hi * 4], u)
        values.append(mz.from_limbs(v.cpu().numpy().view(np.uint64).reshape(1, 4))[0])
    y, carries = sharded.open_carries(values, [hi - lo for lo, hi in spans], u, P_FR)
    parts = [ops.slice_quotient(coef[lo * 4:hi * 4], u, carries[g]) for g, (lo, hi) in enumerate(spans)]
    torch.cuda.synchronize()
    got_q = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
    assert torch.equal(got_q, whole_q)
    assert y == mz.from_limbs(whole_y.cpu().numpy().view(np.uint64).reshape(1, 4))[0]
    c = orc.from_limbs(coef.cpu().numpy().view(np.uint64).reshape(n, 4))
    b = [0] * (n + 1)
    for i in range(n - 1, -1, -1):
        b[i] = (c[i] + u * b[i + 1]) % P_FR
    assert y == b[0] and orc.from_limbs(got_q.cpu().numpy().view(np.uint64).reshape(n, 4)) == b[1:]
    # the driver function at world 1 takes the one-piece recurrence (no exchange) into the caller's buffer: same y, same q
    buf = torch.full((n * 4,), -1, dtype=torch.int64, device=dev)
    y1, q1 = sharded.sharded_open_quotient(sharded.DeviceOpenOps(out=buf), coef, n, u, P_FR, 0, 1)
    torch.cuda.synchronize()
    assert y1 == y and q1.data_ptr() == buf.data_ptr() and torch.equal(q1, got_q)
