"""One transform sharded over W ranks, the GPU compute path (myzkp_amd/sharded.py DeviceOps: mzk_ntt_batch_dev,
mzk_coset_lde_dev, mzk_poly_scale_dev, mzk_ntt_dev on HBM-resident parts) with all ranks inside one process on one GPU
(the exchanges become slicing; tests/test_sharded_ntt.py runs the real all_to_all over gloo): every rank's part must equal,
bit for bit, the corresponding part of the single-GPU transform of the whole vector, itself checked against the oracle
elsewhere (ntt.rs:7-64).  Plus mzk_poly_scale (polynomial.rs:167-174), the twiddle step, against Python integers."""
import ctypes
import numpy as np
import pytest
import orc
from orc import FR, M128

pytestmark = pytest.mark.gpu
LAYOUTS = [("contiguous", "contiguous"), ("contiguous", "cyclic"), ("cyclic", "contiguous")]


@pytest.fixture(scope="module")
def env():
    import torch
    import myzkp_amd as mz
    from myzkp_amd import sharded
    mz.init(0)
    return torch, mz, sharded, mz.lib(), torch.device("cuda", 0), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _parts(torch, x, W, nl, layout):
    v = x.view(-1, nl)
    m = v.shape[0] // W
    if layout == "contiguous":
        return [v[r * m:(r + 1) * m].contiguous().view(-1) for r in range(W)]
    return [v[r::W].contiguous().view(-1) for r in range(W)]


@pytest.mark.parametrize("fid,nl", [(FR, 4), (M128, 2)])
@pytest.mark.parametrize("lg,worlds", [(20, (2, 4, 8)), (12, (1, 2, 8, 64)), (4, (2, 4))])
def test_sharded_ntt_equals_single_gpu_transform(env, fid, nl, lg, worlds):
    torch, mz, sharded, L, dev, st = env
    n = 1 << lg
    w = orc.root_of(fid, lg)
    root = mz.to_limbs([w], nl)
    x = torch.empty(n * nl, dtype=torch.int64, device=dev)
    assert L.mzk_synth_field_dev(fid, ctypes.c_uint64(4400 + lg), ctypes.c_size_t(n), ctypes.c_void_p(x.data_ptr()), st) == 0
    ops = sharded.DeviceOps(fid)
    for inverse in (False, True):
        want = torch.empty_like(x)
        assert L.mzk_ntt_dev(fid, root.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(want.data_ptr()),
                             ctypes.c_size_t(n), int(inverse), st) == 0
        if lg <= 12:
            rc, ref = orc.ntt_fast(fid, w, x.cpu().numpy().view(np.uint64).reshape(-1, nl), inverse)
            assert rc == 0 and np.array_equal(want.cpu().numpy().view(np.uint64).reshape(-1, nl), ref)
        for W in worlds:
            for lin, lout in LAYOUTS:
                outs = sharded.ntt_sharded_simulate([p.clone() for p in _parts(torch, x, W, nl, lin)], orc.MOD[fid], lg, w, ops, inverse, lin, lout)
                torch.cuda.synchronize()
                for r, (got, exp) in enumerate(zip(outs, _parts(torch, want, W, nl, lout))):
                    assert torch.equal(got, exp), (lg, W, lin, lout, inverse, r)


@pytest.mark.parametrize("fid,nl", [(FR, 4), (M128, 2)])
def test_poly_scale_against_python_integers(env, fid, nl):
    torch, mz, sharded, L, dev, st = env
    p = orc.MOD[fid]
    ratio, lead = orc.from_limbs(orc.synth_vector(fid, 91, 2))
    for n in (0, 1, 15, 16, 17, 1000, 4097):
        c = orc.synth_vector(fid, 92 + n, max(n, 1))[:n]
        v = orc.from_limbs(c) if n else []
        assert orc.from_limbs(mz.poly_scale(fid, c, ratio)) == [x * pow(ratio, i, p) % p for i, x in enumerate(v)] if n else mz.poly_scale(fid, c, ratio).shape[0] == 0
        if n:
            assert orc.from_limbs(mz.poly_scale(fid, c, 1, lead)) == [x * lead % p for x in v]
            assert orc.from_limbs(mz.poly_scale(fid, c, ratio, lead)) == [x * lead * pow(ratio, i, p) % p for i, x in enumerate(v)]
    # non-canonical parameters are refused (the ABI's rule for every host parameter)
    bad = np.full((1, nl), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    c = orc.synth_vector(fid, 5, 4)
    out = np.empty_like(c)
    rc = L.mzk_poly_scale(fid, c.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(4), bad.ctypes.data_as(ctypes.c_void_p), None, out.ctypes.data_as(ctypes.c_void_p))
    assert rc == -6   # MZK_E_RANGE


@pytest.mark.parametrize("fid,nl", [(FR, 4), (M128, 2)])
def test_ntt_columns_equals_row_transforms_of_the_transposed_data(env, fid, nl):
    """mzk_ntt_columns_dev (column-major batch of 2..16-point transforms, one trip over the data) against the oracle's
    transform of every column (ntt.rs:7-64), forward and inverse, ragged column counts."""
    torch, mz, sharded, L, dev, st = env
    for W in (2, 4, 8, 16):
        lgw = W.bit_length() - 1
        w = orc.root_of(fid, lgw)
        root = mz.to_limbs([w], nl)
        for cols in (1, 255, 256, 1000):
            x = orc.synth_vector(fid, 300 + W + cols, W * cols)
            xd = torch.from_numpy(x.view(np.int64).reshape(-1).copy()).to(dev)
            yd = torch.empty_like(xd)
            for inverse in (0, 1):
                assert L.mzk_ntt_columns_dev(fid, root.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(yd.data_ptr()),
                                             ctypes.c_size_t(W), ctypes.c_size_t(cols), inverse, st) == 0, L.mzk_last_error().decode()
                torch.cuda.synchronize()
                got = yd.cpu().numpy().view(np.uint64).reshape(W, cols, nl)
                xs = x.reshape(W, cols, nl)
                for c in (0, cols // 2, cols - 1):
                    rc, ref = orc.ntt_fast(fid, w, np.ascontiguousarray(xs[:, c, :]), bool(inverse), 1)
                    assert rc == 0 and np.array_equal(got[:, c, :], ref), (W, cols, inverse, c)
                # all columns at once against the batched row transforms of the transposed data
                rows = np.ascontiguousarray(xs.transpose(1, 0, 2))
                want = mz.ntt_batch(fid, w, rows, bool(inverse)).transpose(1, 0, 2)
                assert np.array_equal(got, want)
    x = torch.zeros(32 * 4 * nl, dtype=torch.int64, device=dev)
    y = torch.zeros_like(x)
    r32 = mz.to_limbs([orc.root_of(fid, 5)], nl).ctypes.data_as(ctypes.c_void_p)
    r4 = mz.to_limbs([orc.root_of(fid, 2)], nl)
    assert L.mzk_ntt_columns_dev(fid, r32, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_size_t(32), ctypes.c_size_t(4), 0, st) == -1
    assert L.mzk_ntt_columns_dev(fid, r4.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_size_t(4), ctypes.c_size_t(4), 0, st) == -1
    assert L.mzk_ntt_columns_dev(fid, r4.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_size_t(3), ctypes.c_size_t(4), 0, st) == -2
    # a root of the wrong order is refused like everywhere else (ntt.rs:15-22)
    assert L.mzk_ntt_columns_dev(fid, r32, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_size_t(4), ctypes.c_size_t(4), 0, st) == -3
