"""Grid-batched commitments and openings of MANY short polynomials against one SRS (mzk_kzg_commit_srs_many_dev,
mzk_kzg_open_srs_many_dev; the _batch forms route to the same pass) -- the reference's call pattern: commit_kzg per row
(das/avail.rs:88-98), per chunk (das/eigenda.rs:92-101), per folded polynomial (algebra/gemini.rs:112-114), open_kzg per
cell (das/avail.rs:132).  Every point / value against the oracle (literal double-and-add MSM for the small cases, its
Pippenger otherwise) AND against the single call."""
import ctypes
import numpy as np
import pytest
import orc
from orc import FR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd as m
    m.init_devices([0])
    yield m
    m.init_devices([0])


def _srs_ex(mz, pts_arr, with_tables):
    """handle with an explicit table choice (0 = none, 1 = default width, 8..22 = that width)"""
    import torch
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    d_p = torch.from_numpy(np.ascontiguousarray(pts_arr).view(np.int64).reshape(-1).copy()).cuda()
    h = ctypes.c_void_p()
    rc = L.mzk_srs_from_device_ex(ctypes.c_void_p(d_p.data_ptr()), ctypes.c_size_t(pts_arr.shape[0]), int(with_tables), ctypes.byref(h), st)
    assert rc == 0, L.mzk_last_error()
    return h


def _commit_many(mz, h, coefs, n=None):
    import torch
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    count = coefs.shape[0]
    n = coefs.shape[1] if n is None else n
    d_c = torch.from_numpy(np.ascontiguousarray(coefs).view(np.int64).reshape(-1).copy()).cuda() if coefs.size else torch.zeros(4, dtype=torch.int64, device="cuda")
    d_o = torch.full((max(count, 1) * 8,), -1, dtype=torch.int64, device="cuda")
    rc = L.mzk_kzg_commit_srs_many_dev(h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), ctypes.c_void_p(d_o.data_ptr()), st)
    assert rc == 0, L.mzk_last_error()
    torch.cuda.synchronize()
    return mz.array_to_points(d_o.cpu().numpy().view(np.uint64).reshape(-1, 8)[:count])


def _commit_one(mz, h, coef):
    import torch
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = coef.shape[0]
    d_c = torch.from_numpy(np.ascontiguousarray(coef).view(np.int64).reshape(-1).copy()).cuda() if n else torch.zeros(4, dtype=torch.int64, device="cuda")
    d_o = torch.full((8,), -1, dtype=torch.int64, device="cuda")
    rc = L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(d_o.data_ptr()), 0, st)
    assert rc == 0, L.mzk_last_error()
    torch.cuda.synchronize()
    return mz.array_to_points(d_o.cpu().numpy().view(np.uint64).reshape(1, 8))[0]


def _skew(coefs, n):
    """rows 0..4 of a batch: all-equal scalars, a bit vector, all zero, r - 1 everywhere, one non-zero coefficient"""
    r = orc.P_FR
    count = coefs.shape[0]
    if count > 0 and n:
        coefs[0][:] = coefs[0][0]
    if count > 1 and n:
        coefs[1][:] = 0
        coefs[1][::2, 0] = 1
    if count > 2:
        coefs[2][:] = 0
    if count > 3 and n:
        coefs[3][:] = orc.to_limbs([r - 1], 4)[0]
    if count > 4 and n:
        coefs[4][:] = 0
        coefs[4][n // 2] = orc.to_limbs([12345], 4)[0]


@pytest.mark.parametrize("n,count,width", [(1, 3, 1), (7, 1, 1), (100, 9, 1), (256, 33, 1), (1024, 6, 1), (1025, 5, 1), (3000, 7, 1), (4096, 3, 1),
                                           (5000, 2, 1), (700, 5, 10), (700, 4, 11), (2100, 3, 12), (1500, 3, 13),
                                           # round 5: from 2^13 coefficients on the pass runs over 12-bit tables the handle builds on first use
                                           (2049, 6, 1), (8191, 5, 1), (8192, 5, 1), (8193, 6, 1), (16384, 6, 1)])
def test_many_commits_equal_oracle_and_single_calls(mz, n, count, width):
    """ragged lengths (not a multiple of the 1024-coefficient chunks, of the lanes, of anything), count = 1, skewed scalar
    rows, infinity among the points, every table width the grid-batched pass supports"""
    p = orc.synth_points(40 + n, n)
    if n > 20:
        p[5] = 0                                    # the point at infinity among the powers
    h = _srs_ex(mz, p, width)
    coefs = np.stack([orc.synth_vector(FR, 9000 + 31 * k + n, n) for k in range(count)])
    _skew(coefs, n)
    got = _commit_many(mz, h, coefs)
    assert len(got) == count
    for k in range(count):
        want = orc.msm_ref(coefs[k], p) if n <= 300 else orc.msm_fast(coefs[k], p)
        assert got[k] == want, (n, count, width, k)
        assert got[k] == _commit_one(mz, h, coefs[k]), (n, k)
    if n >= 8192 and width == 1:
        # the wide tables are the handle's (built once, counted by mzk_srs_table_bytes, used again by the next call; a SHORTER batch
        # against the same handle keeps to its own tables and a longer prefix than 2^13 takes the wide ones): same points either way
        L = mz.lib()
        L.mzk_srs_table_bytes.restype = ctypes.c_size_t
        assert L.mzk_srs_table_bytes(h) == (26 + 22) * n * 64
        assert _commit_many(mz, h, coefs) == got
        flat = coefs.reshape(-1, 4)                  # a batch of SHORTER polynomials out of the same buffer: polynomial k = the k-th run of m coefficients
        for m in (3000, min(n, 8200)):
            assert _commit_many(mz, h, coefs[:2], n=m) == [orc.msm_fast(np.ascontiguousarray(flat[k * m:(k + 1) * m]), p[:m]) for k in range(2)], m
    mz.lib().mzk_srs_free(h)


@pytest.mark.parametrize("n,count,width,bits", [(1024, 7, 1, 248), (1024, 5, 1, 64), (1000, 6, 1, 9), (300, 9, 1, 248), (1024, 5, 10, 248), (700, 5, 12, 130),
                                                (1024, 4, 13, 248), (4096, 3, 1, 248)])
def test_many_commits_of_short_coefficients(mz, n, count, width, bits):
    """the reference's DAS callers commit to 31-byte chunks (das/avail.rs:88-98), so the top windows of every scalar are
    zero and the fixed-capacity entry regions of the one-kernel sort end in a long run of sentinels.  Those are kept out of the last
    bucket's sum (bucket_end, mzk_msm.hip); same points as the oracle and as the single call, whatever the coefficient width."""
    p = orc.synth_points(77 + n + bits, n)
    h = _srs_ex(mz, p, width)
    coefs = np.stack([orc.synth_vector(FR, 4100 + 17 * k + bits, n) for k in range(count)])
    top, rem = bits // 64, bits % 64
    coefs[:, :, top + 1:] = 0
    coefs[:, :, top] &= np.uint64((1 << rem) - 1)
    if count > 2:
        coefs[2][:] = 0                              # a polynomial with no entry at all: its region is sentinels only
        coefs[1][1:] = 0                             # and one with a single coefficient
    got = _commit_many(mz, h, coefs)
    for k in range(count):
        assert got[k] == (orc.msm_ref(coefs[k], p) if n <= 300 else orc.msm_fast(coefs[k], p)), (n, width, bits, k)
        assert got[k] == _commit_one(mz, h, coefs[k]), (n, k)
    mz.lib().mzk_srs_free(h)


@pytest.mark.parametrize("n,count,bits", [(1, 2, 8), (255, 3, 9), (256, 5, 10), (257, 4, 11), (1000, 6, 12), (1024, 5, 10), (2500, 3, 8), (5000, 2, 0)])
def test_many_commits_over_direct_tables(mz, n, count, bits):
    """mzk_srs_build_direct: the same batches with no buckets at all -- every multiple of every window-table row gathered
    directly; ragged lengths around the 256-coefficient workgroups, skewed rows, infinity among the points, every width, the
    width picked by budget (bits = 0), and a prefix of the SRS"""
    import torch
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = orc.synth_points(140 + n, n)
    if n > 20:
        p[5] = 0
    h = _srs_ex(mz, p, 1)
    assert L.mzk_srs_direct_bits(h) == 0
    assert L.mzk_srs_build_direct(h, bits, ctypes.c_size_t(3 << 29 if bits == 0 else 0), st) == 0, L.mzk_last_error()
    got_bits = L.mzk_srs_direct_bits(h)
    assert got_bits == (bits if bits else 8)            # 5000 points, 1.5 GiB: 9 bits would be 2.2 GiB, 8 bits are 1.2 GiB
    coefs = np.stack([orc.synth_vector(FR, 9100 + 31 * k + n, n) for k in range(count)])
    _skew(coefs, n)
    got = _commit_many(mz, h, coefs)
    for k in range(count):
        want = orc.msm_ref(coefs[k], p) if n <= 300 else orc.msm_fast(coefs[k], p)
        assert got[k] == want, (n, bits, k)
    if n > 10:                                          # shorter polynomials against the same handle
        m = n - 7
        got = _commit_many(mz, h, np.ascontiguousarray(coefs[:, :m]))
        for k in range(count):
            assert got[k] == orc.msm_fast(np.ascontiguousarray(coefs[k, :m]), p[:m]), (n, k)
    L.mzk_srs_drop_direct(h)
    assert L.mzk_srs_direct_bits(h) == 0
    got = _commit_many(mz, h, coefs)                    # back on the bucket pass
    for k in range(count):
        assert got[k] == orc.msm_fast(coefs[k], p)
    L.mzk_srs_free(h)


def test_direct_tables_refuse_what_they_cannot_hold(mz):
    import torch
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = orc.synth_points(3, 64)
    h = _srs_ex(mz, p, 1)
    assert L.mzk_srs_build_direct(h, 7, ctypes.c_size_t(0), st) == -1              # width out of range
    assert L.mzk_srs_build_direct(h, 12, ctypes.c_size_t(1 << 20), st) == -1       # 64 x 22 x 2048 x 64 B = 176 MiB > 1 MiB
    assert b"budget" in L.mzk_last_error()
    assert L.mzk_srs_direct_bits(h) == 0
    L.mzk_srs_free(h)
    big = orc.synth_points(4, (1 << 14) + 1)
    h = _srs_ex(mz, big, 0)
    assert L.mzk_srs_build_direct(h, 8, ctypes.c_size_t(0), st) == -1              # more than 2^14 points
    L.mzk_srs_free(h)


def test_many_commits_prefix_of_a_longer_srs_and_empty_cases(mz):
    """n below the SRS length (commit_kzg of a shorter polynomial), n = 0 (every point infinity), count = 0 (nothing written),
    n above the SRS length (the reference's index panic -> MZK_E_LENGTH)"""
    import torch
    N = 2000
    p = orc.synth_points(77, N)
    h = _srs_ex(mz, p, 1)
    for n in (1, 999, 1500):
        coefs = np.stack([orc.synth_vector(FR, 600 + k + n, n) for k in range(4)])
        got = _commit_many(mz, h, coefs)
        for k in range(4):
            assert got[k] == orc.msm_fast(coefs[k], p[:n])
    got = _commit_many(mz, h, np.zeros((5, 0, 4), dtype=np.uint64), n=0)
    assert got == [orc.msm_ref(np.zeros((0, 4), dtype=np.uint64), p[:0])] * 5
    assert _commit_many(mz, h, np.zeros((0, 8, 4), dtype=np.uint64), n=8) == []
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = torch.zeros(64, dtype=torch.int64, device="cuda")
    rc = L.mzk_kzg_commit_srs_many_dev(h, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(N + 1), ctypes.c_size_t(2), ctypes.c_void_p(d.data_ptr()), st)
    assert rc == -5 and b"index out of bounds" in L.mzk_last_error()
    L.mzk_srs_free(h)


def test_many_commits_on_a_handle_without_narrow_tables_take_the_lanes(mz):
    """a handle with 16-bit tables or with no tables has no grid-batched pass: the call still answers (one commit per lane)"""
    n, count = 600, 3
    p = orc.synth_points(5, n)
    coefs = np.stack([orc.synth_vector(FR, 70 + k, n) for k in range(count)])
    for width in (0, 16):
        h = _srs_ex(mz, p, width)
        got = _commit_many(mz, h, coefs)
        for k in range(count):
            assert got[k] == orc.msm_fast(coefs[k], p)
        mz.lib().mzk_srs_free(h)


def test_batch_entry_points_route_to_the_same_pass(mz):
    """Srs.commit_batch (mzk_kzg_commit_srs_batch, host buffers) with ONE context: 40 polynomials of 2^10 coefficients"""
    n, count = 1 << 10, 40
    p = orc.synth_points(11, n)
    h = mz.Srs(p)
    coefs = np.stack([orc.synth_vector(FR, 3000 + k, n) for k in range(count)])
    got = h.commit_batch(coefs)
    for k in range(count):
        assert got[k] == orc.msm_fast(coefs[k], p), k
    h.close()


@pytest.mark.parametrize("n,count", [(1, 3), (2, 2), (257, 5), (1024, 9), (1025, 4), (4097, 3), (16384, 2)])
def test_many_openings_equal_definition_and_single_calls(mz, n, count):
    """open_kzg (kzg.rs:61-72) per polynomial at its own point: y = f(u) by the oracle's Horner, w = the oracle's witness
    (literal kzg_open_ref for the small cases), and both equal to mzk_kzg_open_srs_dev"""
    import torch
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = orc.synth_points(300 + n, n)
    h = mz.Srs(p)
    coefs = np.stack([orc.synth_vector(FR, 2000 + 7 * k + n, n) for k in range(count)])
    us = orc.synth_vector(FR, 4000 + n, count)
    us[1] = us[0]                                     # two polynomials at one point
    if count > 2:
        us[2] = 0                                     # u = 0: y = c_0
    d_c = torch.from_numpy(coefs.view(np.int64).reshape(-1).copy()).cuda()
    d_y = torch.full((count * 4,), -1, dtype=torch.int64, device="cuda")
    d_w = torch.full((count * 8,), -1, dtype=torch.int64, device="cuda")
    rc = L.mzk_kzg_open_srs_many_dev(h._h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), us.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.c_void_p(d_y.data_ptr()), ctypes.c_void_p(d_w.data_ptr()), st)
    assert rc == 0, L.mzk_last_error()
    torch.cuda.synchronize()
    ys = orc.from_limbs(d_y.cpu().numpy().view(np.uint64).reshape(count, 4))
    ws = mz.array_to_points(d_w.cpu().numpy().view(np.uint64).reshape(count, 8))
    y1 = torch.zeros(4, dtype=torch.int64, device="cuda"); w1 = torch.zeros(8, dtype=torch.int64, device="cuda")
    for k in range(count):
        u = orc.from_limbs(us[k:k + 1])[0]
        assert ys[k] == orc.poly_eval(FR, coefs[k], u), (n, k)
        if n <= 257:
            y_ref, w_ref = orc.kzg_open_ref(coefs[k], u, p)
            assert (ys[k], ws[k]) == (y_ref, w_ref), (n, k)
        assert L.mzk_kzg_open_srs_dev(h._h, ctypes.c_void_p(d_c.data_ptr() + k * n * 32), ctypes.c_size_t(n), us[k].ctypes.data_as(ctypes.c_void_p),
                                      ctypes.c_void_p(y1.data_ptr()), ctypes.c_void_p(w1.data_ptr()), st) == 0
        torch.cuda.synchronize()
        assert orc.from_limbs(y1.cpu().numpy().view(np.uint64).reshape(1, 4))[0] == ys[k]
        assert mz.array_to_points(w1.cpu().numpy().view(np.uint64).reshape(1, 8))[0] == ws[k], (n, k)
    if n <= 4097:                                    # the same openings over direct tables
        assert L.mzk_srs_build_direct(h._h, 8, ctypes.c_size_t(0), st) == 0, L.mzk_last_error()
        d_y2 = torch.full((count * 4,), -1, dtype=torch.int64, device="cuda"); d_w2 = torch.full((count * 8,), -1, dtype=torch.int64, device="cuda")
        assert L.mzk_kzg_open_srs_many_dev(h._h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), us.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_void_p(d_y2.data_ptr()), ctypes.c_void_p(d_w2.data_ptr()), st) == 0, L.mzk_last_error()
        torch.cuda.synchronize()
        assert torch.equal(d_y, d_y2) and torch.equal(d_w, d_w2)
    # a non-canonical point is rejected before anything is enqueued (kzg_open's MZK_E_RANGE)
    bad = us.copy(); bad[count - 1] = np.array([2**64 - 1] * 4, dtype=np.uint64)
    rc = L.mzk_kzg_open_srs_many_dev(h._h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), bad.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.c_void_p(d_y.data_ptr()), ctypes.c_void_p(d_w.data_ptr()), st)
    assert rc == -6
    h.close()


def test_many_commits_256_by_1024_the_das_shape(mz):
    """the size VERDICT r03 names: 256 rows of 2^10 coefficients against one SRS, every row against the oracle's Pippenger"""
    n, count = 1 << 10, 256
    p = orc.synth_points(2024, n)
    h = _srs_ex(mz, p, 1)
    coefs = orc.synth_vector(FR, 555, n * count).reshape(count, n, 4)
    got = _commit_many(mz, h, coefs)
    for k in range(count):
        assert got[k] == orc.msm_fast(coefs[k], p), k
    mz.lib().mzk_srs_free(h)


def test_many_commits_more_polynomials_than_one_pass_holds(mz):
    """the pass handles at most 2^21 buckets / 2^22 coefficients (bucket form) or 2^12 polynomials (direct form) at a time and
    loops beyond that: 258 polynomials of 2^14 coefficients (two passes of the bucket form), 16500 polynomials of 8 coefficients
    over direct tables (five passes of 2^12) and over the bucket form (two), and openings across the same boundary"""
    import torch
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # bucket form: 256 + 2 polynomials of 2^14 coefficients
    n, count = 1 << 14, 258
    p = orc.synth_points(31, n)
    h = _srs_ex(mz, p, 1)
    coefs = orc.synth_vector(FR, 32, n * count).reshape(count, n, 4)
    got = _commit_many(mz, h, coefs)
    for k in (0, 1, 127, 255, 256, 257):
        assert got[k] == orc.msm_fast(coefs[k], p), k
    # every point against the single call (device compare: the oracle would take minutes)
    d_c = torch.from_numpy(coefs.view(np.int64).reshape(-1).copy()).cuda()
    d_many = torch.zeros(count * 8, dtype=torch.int64, device="cuda")
    d_one = torch.zeros(count * 8, dtype=torch.int64, device="cuda")
    assert L.mzk_kzg_commit_srs_many_dev(h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), ctypes.c_void_p(d_many.data_ptr()), st) == 0
    for k in range(count):
        assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(d_c.data_ptr() + k * n * 32), ctypes.c_size_t(n), ctypes.c_void_p(d_one.data_ptr() + k * 64), 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(d_many, d_one)
    # openings across the pass boundary (2^22 / 2^14 = 256 polynomials per pass)
    us = orc.synth_vector(FR, 33, count)
    d_y = torch.zeros(count * 4, dtype=torch.int64, device="cuda"); d_w = torch.zeros(count * 8, dtype=torch.int64, device="cuda")
    assert L.mzk_kzg_open_srs_many_dev(h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), us.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_void_p(d_y.data_ptr()), ctypes.c_void_p(d_w.data_ptr()), st) == 0, L.mzk_last_error()
    torch.cuda.synchronize()
    ys = orc.from_limbs(d_y.cpu().numpy().view(np.uint64).reshape(count, 4))
    y1 = torch.zeros(4, dtype=torch.int64, device="cuda"); w1 = torch.zeros(8, dtype=torch.int64, device="cuda")
    for k in (0, 255, 256, 257):
        assert ys[k] == orc.poly_eval(FR, coefs[k], orc.from_limbs(us[k:k + 1])[0]), k
        assert L.mzk_kzg_open_srs_dev(h, ctypes.c_void_p(d_c.data_ptr() + k * n * 32), ctypes.c_size_t(n), us[k].ctypes.data_as(ctypes.c_void_p),
                                      ctypes.c_void_p(y1.data_ptr()), ctypes.c_void_p(w1.data_ptr()), st) == 0
        torch.cuda.synchronize()
        assert torch.equal(w1, d_w[8 * k:8 * k + 8]) and torch.equal(y1, d_y[4 * k:4 * k + 4]), k
    L.mzk_srs_free(h)
    del d_c, d_many, d_one
    # direct form: 4 x 4096 + 116 polynomials of 8 coefficients
    n, count = 8, 16500
    p = orc.synth_points(41, n)
    h = _srs_ex(mz, p, 1)
    assert L.mzk_srs_build_direct(h, 10, ctypes.c_size_t(0), st) == 0, L.mzk_last_error()
    coefs = orc.synth_vector(FR, 42, n * count).reshape(count, n, 4)
    got = _commit_many(mz, h, coefs)
    for k in list(range(0, count, 997)) + [4095, 4096, 4097, 16383, 16384, 16385, count - 1]:
        assert got[k] == orc.msm_ref(coefs[k], p), k
    L.mzk_srs_drop_direct(h)
    assert _commit_many(mz, h, coefs) == got            # the bucket form agrees on all 16500
    L.mzk_srs_free(h)
