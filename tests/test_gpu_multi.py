"""Multi-GPU MSM / KZG commit INSIDE the C ABI (mzk_init_devices + the *_multi entry points; SURVEY 8e, BASELINE
configs[3]) -- one process, W contexts.  Contexts may share a device ordinal, so the whole path (W streams, W
workspaces and table caches, pinned gather of the 128-byte partials, fold on context 0) runs on a one-GPU box; the
last case uses every visible device."""
import numpy as np
import pytest
import orc
from orc import FR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd as m
    yield m
    m.init_devices([0])   # back to ONE context (mzk_init(0) would keep the extra contexts: it is idempotent)


def _inputs(n, seed):
    s = orc.synth_vector(FR, seed, n)
    p = orc.synth_points(seed + 1, n)
    if n > 20:
        p[17] = 0
        s[3] = 0
    return s, p


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_msm_multi_equals_oracle(mz, world):
    mz.init_devices([0] * world)
    assert mz.ctx_count() == world
    for n in (0, 1, world - 1, 5000):
        s, p = _inputs(n, 700 + world)
        assert mz.msm_g1_multi(s, p) == orc.msm_fast(s, p), (world, n)
    # context 0 still serves the single-device entry points, and so does any other context once selected
    s, p = _inputs(300, 77)
    want = orc.msm_fast(s, p)
    assert mz.msm_g1(s, p) == want
    mz.ctx_select(world - 1)
    assert mz.msm_g1(s, p) == want
    mz.ctx_select(0)


def test_shard_range_matches_the_python_sharding(mz):
    from myzkp_amd import sharded
    for n in (0, 1, 7, 1000, (1 << 24) + 5):
        for world in (1, 2, 3, 8):
            assert [mz.shard_range(n, r, world) for r in range(world)] == [sharded.shard_range(n, r, world) for r in range(world)]


@pytest.mark.parametrize("world", [2, 5])
def test_srs_multi_commit_host_and_device_coefficients(mz, world):
    import torch
    mz.init_devices([0] * world)
    n = (1 << 15) + 11                   # world 2: shards above the window-table threshold (2^14), world 5: below it
    s, p = _inputs(n, 800 + world)
    h = mz.SrsMulti(powers=p)
    assert h.world == world and h.lo[0] == 0 and h.lo[-1] == n
    assert h.commit(s) == orc.msm_fast(s, p)
    m = n - 4097                         # shorter polynomial: the last shards see fewer (or no) coefficients
    assert h.commit(s[:m]) == orc.msm_fast(s[:m], p[:m])
    assert h.commit(s[:5]) == orc.msm_fast(s[:5], p[:5])
    with pytest.raises(mz.MzkError) as e:
        h.commit(np.concatenate([s, s[:1]]))
    assert e.value.code == -5            # powers[i] out of bounds, polynomial.rs:162
    # device-resident shards (all contexts share cuda:0 here)
    dev = torch.device("cuda", 0)
    shards = [torch.from_numpy(s[h.lo[r]:h.lo[r + 1]].view(np.int64).reshape(-1).copy()).to(dev) for r in range(world)]
    torch.cuda.synchronize()
    assert h.commit_dev([t.data_ptr() for t in shards], n) == orc.msm_fast(s, p)
    h.close()


def test_setup_srs_multi_then_commit_is_the_trapdoor_identity(mz):
    world = 3
    mz.init_devices([0] * world)
    n = (1 << 16) + 3
    alpha = orc.from_limbs(orc.synth_vector(FR, 901, 1))[0]
    s = orc.synth_vector(FR, 902, n)
    want = orc.ec_mul(0, (1, 2), orc.poly_eval(FR, s, alpha))
    for with_tables in (0, 1):
        h = mz.SrsMulti(alpha=alpha, max_d=n - 1, with_tables=with_tables)
        assert h.commit(s) == want, with_tables
        h.close()


def test_every_visible_device(mz):
    import torch
    count = torch.cuda.device_count()
    mz.init_devices(list(range(count)))
    s, p = _inputs(20000, 990)
    assert mz.msm_g1_multi(s, p) == orc.msm_fast(s, p)
    alpha = 0x1234567
    n = 1 << 14
    h = mz.SrsMulti(alpha=alpha, max_d=n - 1)
    c = orc.synth_vector(FR, 991, n)
    assert h.commit(c) == orc.ec_mul(0, (1, 2), orc.poly_eval(FR, c, alpha))
    h.close()


def test_srs_handle_is_shared_by_contexts_on_the_same_device(mz):
    """A handle is plain device memory: a second context on the SAME GPU (own stream + workspace) may commit against it --
    that is how two commits are kept in flight (bench.py `kzg_commit_two_in_flight`).  Only a context on another
    device is refused (MZK_E_ARG; needs two GPUs to provoke)."""
    import torch
    mz.init_devices([0, 0])
    for n in (64, (1 << 14) + 5):
        s, p = _inputs(n, 55 + n)
        want = orc.msm_fast(s, p)
        h = mz.Srs(p)                        # built on context 0
        mz.ctx_select(1)
        assert h.commit(s) == want
        mz.ctx_select(0)
        assert h.commit(s) == want
        if torch.cuda.device_count() > 1:
            mz.init_devices([0, 1])
            mz.ctx_select(1)
            with pytest.raises(mz.MzkError) as e:
                h.commit(s)
            assert e.value.code == -1
            mz.init_devices([0, 0])
        h.close()


@pytest.mark.parametrize("nctx", [1, 2, 4])
def test_commit_batch_equals_single_commits(mz, nctx):
    """mzk_kzg_commit_srs_batch: `count` commit_kzg calls (kzg.rs:57-59) with one commit in flight per context of the
    GPU -- every point equal to the oracle's and to the one-at-a-time result, for ragged lane assignments (count not a
    multiple of the contexts), the small path (n < 4096) and the general one, and count = 0."""
    mz.init_devices([0] * nctx)
    for n, count in ((300, 5), ((1 << 13) + 3, 7), (1 << 12, 1)):
        p = orc.synth_points(900 + n, n)
        h = mz.Srs(p)
        coefs = np.stack([orc.synth_vector(FR, 1000 + 17 * k + n, n) for k in range(count)])
        coefs[0][1] = 0
        got = h.commit_batch(coefs)
        assert len(got) == count
        for k in range(count):
            assert got[k] == orc.msm_fast(coefs[k], p), (n, k)
            assert got[k] == h.commit(coefs[k])
        assert h.commit_batch(np.zeros((0, n, 4), dtype=np.uint64)) == []
        h.close()


def test_commit_batch_dev_forks_from_and_joins_the_callers_stream(mz):
    """The device form on a caller stream that is not a context stream: inputs produced on that stream just before the
    call, results consumed on it right after -- no host synchronisation in between."""
    import ctypes, torch
    mz.init_devices([0, 0, 0])
    L = mz.lib()
    n, count = 1 << 12, 6
    p = orc.synth_points(77, n)
    h = mz.Srs(p)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sp = ctypes.c_void_p(st.cuda_stream)
        d_c = torch.empty(count * n * 4, dtype=torch.int64, device="cuda")
        for k in range(count):      # synthetic coefficients written on the caller's stream
            assert L.mzk_synth_field_dev(0, ctypes.c_uint64(500 + k), ctypes.c_size_t(n), ctypes.c_void_p(d_c.data_ptr() + k * n * 32), sp) == 0
        d_o = torch.zeros(count * 8, dtype=torch.int64, device="cuda")
        rc = L.mzk_kzg_commit_srs_batch_dev(h._h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count),
                                            ctypes.c_void_p(d_o.data_ptr()), ctypes.c_int(0), sp)
        assert rc == 0, L.mzk_last_error()
        res = d_o.clone()           # enqueued on the caller's stream behind the join
    st.synchronize()
    got = mz.array_to_points(res.cpu().numpy().view(np.uint64).reshape(count, 8))
    for k in range(count):
        assert got[k] == orc.msm_fast(orc.synth_vector(FR, 500 + k, n), p)
    h.close()


@pytest.mark.parametrize("nctx", [1, 3])
def test_open_batch_equals_single_openings(mz, nctx):
    """mzk_kzg_open_srs_batch_dev: open_kzg (kzg.rs:61-72) of several polynomials, each at its own point, one opening in
    flight per context -- every (y, w) equal to the single device call and to the definition: y = f(u) by the oracle's
    Horner, w = the commitment of (f - y) / (X - u)."""
    import ctypes, torch
    mz.init_devices([0] * nctx)
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n, count in ((257, 5), ((1 << 12) + 1, 4)):
        p = orc.synth_points(300 + n, n)
        h = mz.Srs(p)
        coefs = np.stack([orc.synth_vector(FR, 2000 + 7 * k + n, n) for k in range(count)])
        us = orc.synth_vector(FR, 4000 + n, count)
        us[1] = us[0]                                     # two polynomials at one point
        d_c = torch.from_numpy(coefs.view(np.int64).reshape(-1).copy()).cuda()
        d_y = torch.zeros(count * 4, dtype=torch.int64, device="cuda")
        d_w = torch.zeros(count * 8, dtype=torch.int64, device="cuda")
        rc = L.mzk_kzg_open_srs_batch_dev(h._h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count),
                                          us.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_y.data_ptr()), ctypes.c_void_p(d_w.data_ptr()), ctypes.c_int(0), st)
        assert rc == 0, L.mzk_last_error()
        torch.cuda.synchronize()
        ys = orc.from_limbs(d_y.cpu().numpy().view(np.uint64).reshape(count, 4))
        ws = mz.array_to_points(d_w.cpu().numpy().view(np.uint64).reshape(count, 8))
        y1 = torch.zeros(4, dtype=torch.int64, device="cuda"); w1 = torch.zeros(8, dtype=torch.int64, device="cuda")
        for k in range(count):
            u = orc.from_limbs(us[k:k + 1])[0]
            assert ys[k] == orc.poly_eval(FR, coefs[k], u), (n, k)
            assert L.mzk_kzg_open_srs_dev(h._h, ctypes.c_void_p(d_c.data_ptr() + k * n * 32), ctypes.c_size_t(n), us[k].ctypes.data_as(ctypes.c_void_p),
                                          ctypes.c_void_p(y1.data_ptr()), ctypes.c_void_p(w1.data_ptr()), st) == 0
            torch.cuda.synchronize()
            assert orc.from_limbs(y1.cpu().numpy().view(np.uint64).reshape(1, 4))[0] == ys[k]
            assert mz.array_to_points(w1.cpu().numpy().view(np.uint64).reshape(1, 8))[0] == ws[k]
        h.close()


# ---- one transform sharded over the contexts (mzk_ntt_multi[_dev]; SURVEY 8e four-step layout) ------------------------------
@pytest.mark.parametrize("world", [1, 2, 4, 8, 16])
def test_ntt_multi_host_vector_equals_oracle(mz, world):
    mz.init_devices([0] * world)
    for fid in (orc.FR, orc.M128):
        for lg in (8, 13):
            w = orc.root_of(fid, lg)
            x = orc.synth_vector(fid, 900 + lg, 1 << lg)
            for inverse in (False, True):
                rc, want = orc.ntt_fast(fid, w, x, inverse)
                assert rc == 0 and np.array_equal(mz.ntt_multi(fid, w, x, inverse), want), (world, fid, lg, inverse)


@pytest.mark.parametrize("world", [2, 8])
def test_ntt_multi_dev_layouts_equal_single_gpu_transform(mz, world):
    """Parts resident in HBM, the three layout pairs, forward and inverse, 2^20 points: every part bit-identical to the
    corresponding part of the single-context transform; the inputs are left untouched."""
    import ctypes, torch
    mz.init_devices([0] * world)
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lg = 20
    n = 1 << lg
    for fid, nl in ((orc.FR, 4), (orc.M128, 2)):
        w = orc.root_of(fid, lg)
        root = mz.to_limbs([w], nl)
        x = torch.empty(n * nl, dtype=torch.int64, device=dev)
        assert L.mzk_synth_field_dev(fid, ctypes.c_uint64(4242), ctypes.c_size_t(n), ctypes.c_void_p(x.data_ptr()), st) == 0
        for inverse in (False, True):
            want = torch.empty_like(x)
            assert L.mzk_ntt_dev(fid, root.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(want.data_ptr()),
                                 ctypes.c_size_t(n), int(inverse), st) == 0
            torch.cuda.synchronize()

            def parts(t, layout):
                v = t.view(-1, nl)
                m = n // world
                if layout == mz.LAYOUT_CONTIGUOUS:
                    return [v[r * m:(r + 1) * m].contiguous().view(-1) for r in range(world)]
                return [v[r::world].contiguous().view(-1) for r in range(world)]

            for lin, lout in ((0, 0), (0, 1), (1, 0)):
                ins = parts(x, lin)
                keep = [p.clone() for p in ins]
                outs = [torch.zeros_like(p) for p in ins]
                torch.cuda.synchronize()
                mz.ntt_multi_dev(fid, w, [p.data_ptr() for p in ins], [p.data_ptr() for p in outs], n, inverse, lin, lout)
                for r, (got, exp) in enumerate(zip(outs, parts(want, lout))):
                    assert torch.equal(got, exp), (world, fid, inverse, lin, lout, r)
                assert all(torch.equal(a, b) for a, b in zip(ins, keep))


def test_ntt_multi_argument_errors(mz):
    import ctypes
    L = mz.lib()
    mz.init_devices([0] * 3)
    x = orc.synth_vector(orc.FR, 1, 9)
    with pytest.raises(mz.MzkError) as e:
        mz.ntt_multi(orc.FR, orc.root_of(orc.FR, 3), x[:9])
    assert e.value.code in (-1, -2)
    mz.init_devices([0] * 4)
    with pytest.raises(mz.MzkError) as e:        # world^2 > n
        mz.ntt_multi(orc.FR, orc.root_of(orc.FR, 3), x[:8])
    assert e.value.code == -1
    x = orc.synth_vector(orc.FR, 1, 64)
    with pytest.raises(mz.MzkError) as e:        # root of the wrong order (ntt.rs:15-18)
        mz.ntt_multi(orc.FR, orc.root_of(orc.FR, 7), x)
    assert e.value.code == -3
    with pytest.raises(mz.MzkError) as e:        # not primitive (ntt.rs:19-22)
        mz.ntt_multi(orc.FR, orc.root_of(orc.FR, 5), x)
    assert e.value.code == -4
    with pytest.raises(mz.MzkError) as e:        # cyclic -> cyclic is not offered
        mz.ntt_multi_dev(orc.FR, orc.root_of(orc.FR, 6), [8, 8, 8, 8], [8, 8, 8, 8], 64, False, 1, 1)
    assert e.value.code == -1


def test_srs_multi_shards_of_exactly_4096_points(mz):
    """8 contexts x 4096 points: every shard is a commit of exactly 4096 coefficients, which takes the three-launch small path
    (10-bit tables) and writes an XYZZ partial record instead of the affine point."""
    mz.init_devices([0] * 8)
    n = 8 * 4096
    s, p = _inputs(n, 4242)
    h = mz.SrsMulti(powers=p)
    assert [h.lo[r + 1] - h.lo[r] for r in range(8)] == [4096] * 8
    assert h.commit(s) == orc.msm_fast(s, p)
    assert h.commit(s[:n - 5]) == orc.msm_fast(s[:n - 5], p[:n - 5])       # last shard short by five: 4091 coefficients
    h.close()


def test_merkle_handles_follow_their_context(mz):
    """a tree handle records its owning context.  Opening it while ANOTHER context is current enters the owner
    (its stream, its workspace); one open_multi call refuses trees of two contexts; a handle survives re-initialisation on
    the same device (its build stream is gone, its memory is not)."""
    fid, n = orc.M128, 1 << 10
    cw = orc.synth_vector(fid, 4242, n)
    want_root = orc.merkle_commit_field_ref(fid, cw)
    mz.init_devices([0, 0])
    t0 = mz.MerkleTree(fid, elems=cw)
    mz.ctx_select(1)
    t1 = mz.MerkleTree(fid, elems=cw)
    idx = [0, 1, 513, n - 1]
    try:
        # current context 1, tree of context 0 (and the other way round)
        assert t0.root() == want_root and t1.root() == want_root
        p0 = t0.open_many(idx)
        mz.ctx_select(0)
        p1 = t1.open_many(idx)
        assert p0 == p1
        for i, path in zip(idx, p0):
            assert orc.merkle_verify_ref(want_root, i, path, orc.bincode_field(orc.from_limbs(cw[i:i + 1])[0], 2)), i
        assert np.array_equal(t1.leaves(idx), cw[idx])
        with pytest.raises(mz.MzkError) as e:
            mz.merkle_open_multi([t0, t1], [idx, idx])
        assert e.value.code == -1
        assert mz.merkle_open_multi([t0, t0], [idx, idx[:2]]) == [p0, p0[:2]]
        # re-initialise: one context on the same device.  t0's context index still exists there, t1's does not.
        mz.init_devices([0])
        assert t0.root() == want_root and t0.open_many(idx) == p0
        with pytest.raises(mz.MzkError) as e:
            t1.open_many(idx)
        assert e.value.code == -1
    finally:
        t0.close(); t1.close()
