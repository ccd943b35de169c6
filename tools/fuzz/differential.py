"""Randomised differential run of the C ABI against the oracle, with random shapes -- sizes that are not powers of two where the
entry point takes them, ragged batches, empty inputs, special scalars.  A longer-running companion of tests/ (which pins fixed
seeds and sizes): every case prints nothing unless it differs; the summary line counts the cases per entry point.

usage: differential.py [seconds=60] [seed=1]
tests/test_gpu_fuzz_slice.py runs a 20-second seeded slice of it under `pytest -m gpu` (run(budget, seed)).
The oracle (tests/orc.py) is the checker, as in tests/."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import myzkp_amd as mz
import orc
from orc import FR, M128

rng = random.Random(1)
counts, failures = {}, []
NL = {FR: 4, M128: 2}


def vec(fid, n):
    return orc.synth_vector(fid, rng.getrandbits(40), max(n, 1))[:n]


def special_scalars(n):
    s = orc.synth_vector(FR, rng.getrandbits(40), max(n, 1))[:n].copy()
    p = orc.P_FR
    pool = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 253, (1 << 128) - 1, 1 << 127, 0xFFFF, 0x8000, 0x10000]
    mode = rng.randrange(6)
    if mode >= 4 and n:          # short scalars (witness values): bits, bytes, 16 / 64 / 128 / 248-bit -- uneven sort bins, short segments
        bits = rng.choice([1, 8, 16, 64, 128, 248])
        full, rem = bits // 64, bits % 64
        s[:, full + (1 if rem else 0):] = 0
        if rem:
            s[:, full] &= np.uint64((1 << rem) - 1)
        if mode == 5:
            s[rng.randrange(n):] = 0      # and a zero tail
        return s
    for i in range(n):
        if mode == 1 and rng.random() < 0.3 or mode == 2:
            s[i] = orc.to_limbs([rng.choice(pool)], 4)[0]
        elif mode == 3:
            s[i] = s[0]
    return s


def check(name, ok, info):
    counts[name] = counts.get(name, 0) + 1
    if not ok:
        failures.append((name, info))
        print("DIFF", name, info, flush=True)


def case_ntt():
    fid = rng.choice((FR, M128))
    lg = rng.choice([0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21])      # 2^20: the large-tile geometry; odd and even level counts
    n = 1 << lg
    w = orc.root_of(fid, lg) if lg else 1
    x = vec(fid, n)
    inv = rng.random() < 0.5
    rc, want = orc.ntt_fast(fid, w, x, inv)
    check("ntt", rc == 0 and np.array_equal(mz.ntt(fid, w, x, inv), want), (fid, lg, inv))


def case_ntt_batch():
    fid = rng.choice((FR, M128))
    lg, batch = rng.choice([1, 2, 4, 7, 9, 10, 12, 14]), rng.choice([1, 2, 3, 5, 17])
    n = 1 << lg
    w = orc.root_of(fid, lg)
    x = vec(fid, n * batch).reshape(batch, n, NL[fid])
    inv = rng.random() < 0.5
    got = mz.ntt_batch(fid, w, x, inv)
    ok = all(np.array_equal(got[b], orc.ntt_fast(fid, w, np.ascontiguousarray(x[b]), inv)[1]) for b in range(batch))
    check("ntt_batch", ok, (fid, lg, batch, inv))


def case_lde():
    fid = rng.choice((FR, M128))
    lgo = rng.choice([1, 3, 6, 9, 11, 12, 14, 16])
    order = 1 << lgo
    n_coef = rng.choice([0, 1, order, order // 2, max(1, order // 4), rng.randrange(0, order + 1)])
    p = orc.MOD[fid]
    offset, g = rng.randrange(1, p), orc.root_of(fid, lgo)
    c = vec(fid, n_coef)
    # closed form through the oracle's transform: ntt(g, pad(coef[i] * offset^i))
    sc = orc.to_limbs([int(v) * pow(offset, i, p) % p for i, v in enumerate(orc.from_limbs(c))] + [0] * (order - n_coef), NL[fid]) if order else c
    rc, want = orc.ntt_fast(fid, g, sc, False)
    check("coset_lde", rc == 0 and np.array_equal(mz.coset_lde(fid, c, offset, g, order), want), (fid, lgo, n_coef))
    if rng.random() < 0.3 and n_coef:
        batch = rng.choice([2, 3])
        cs = vec(fid, n_coef * batch).reshape(batch, n_coef, NL[fid])
        got = mz.coset_lde_batch(fid, cs, offset, g, order)
        ok = all(np.array_equal(got[b], mz.coset_lde(fid, np.ascontiguousarray(cs[b]), offset, g, order)) for b in range(batch))
        check("coset_lde_batch", ok, (fid, lgo, n_coef, batch))


def case_scale_columns():
    fid = rng.choice((FR, M128))
    p = orc.MOD[fid]
    n = rng.choice([1, 5, 16, 17, 255, 1000, 4097])
    c = vec(fid, n)
    ratio, lead = rng.randrange(p), rng.randrange(p)
    want = [int(v) * lead * pow(ratio, i, p) % p for i, v in enumerate(orc.from_limbs(c))]
    check("poly_scale", orc.from_limbs(mz.poly_scale(fid, c, ratio, lead)) == want, (fid, n))


def case_msm():
    n = rng.choice([0, 1, 2, 3, 17, 100, 1000, 4095, 4096, 4097, 9000, 20000, rng.randrange(1, 30000), rng.randrange(1 << 15, 1 << 17)])
    if rng.random() < 0.04:
        n = rng.randrange(1 << 18, (1 << 18) + (1 << 16))                 # from 2^18 pairs on the host-buffer calls upload and compute in pieces (msm_chunked_impl)
    s = special_scalars(n)
    pts = orc.synth_points(rng.getrandbits(40), max(n, 1))[:n].copy()
    if n > 4:
        for _ in range(rng.randrange(0, 4)):
            pts[rng.randrange(n)] = 0                                     # points at infinity
        if rng.random() < 0.3:
            pts[1] = pts[0]                                               # P next to P
        if rng.random() < 0.3:
            q = orc.arr_to_pts(pts[2:3])[0]
            if q != (0, 0):
                pts[3] = orc.pts_to_arr([(q[0], orc.P_FQ - q[1])])[0]     # P next to -P
                s[3] = s[2]
    want = orc.msm_fast(s, pts)
    check("msm_g1", mz.msm_g1(s, pts) == want, n)
    if n:
        h = mz.Srs(pts)
        check("commit_srs", h.commit(s) == want, n)
        m = rng.randrange(0, n + 1)
        check("commit_srs_prefix", h.commit(s[:m]) == orc.msm_fast(s[:m], pts[:m]), (n, m))
        h.close()


def case_many():
    """grid-batched commitments (bucket pass, then direct tables of a random width) and a table budget that degrades the layout"""
    import ctypes
    L = mz.lib()
    n = rng.choice([1, 2, 31, 256, 257, 1000, 1024, 1025, 2049, 4096, rng.randrange(1, 6000)])
    count = rng.choice([1, 2, 3, 4, 5, 9, 17, 40])
    pts = orc.synth_points(rng.getrandbits(40), n).copy()
    if n > 4 and rng.random() < 0.5:
        pts[rng.randrange(n)] = 0
    rows = np.stack([special_scalars(n) for _ in range(count)])
    h = mz.Srs(pts)
    m = rng.choice([n, n, rng.randrange(0, n + 1)])
    want = [orc.msm_fast(np.ascontiguousarray(rows[k, :m]), pts[:m]) for k in range(count)]
    check("commit_many_buckets", h.commit_many(np.ascontiguousarray(rows[:, :m])) == want, (n, m, count))
    bits = rng.choice([8, 9, 10, 11, 12])
    if n * (254 // bits + 1) * (1 << (bits - 1)) * 64 <= (3 << 29):
        h.build_direct(bits)
        check("commit_many_direct", h.commit_many(np.ascontiguousarray(rows[:, :m])) == want, (n, m, count, bits))
    h.close()
    big = rng.choice([1 << 15, (1 << 15) + rng.randrange(1, 5000)])
    bp = orc.synth_points(rng.getrandbits(40), big)
    bs = special_scalars(big)
    L.mzk_set_table_budget(ctypes.c_size_t(rng.choice([1, 3, 5, 9]) * big * 64))
    try:
        hb = mz.Srs(bp)
        check("commit_under_table_budget", hb.commit(bs) == orc.msm_fast(bs, bp), (big, L.mzk_srs_bucket_sets(hb._h)))
        hb.close()
    finally:
        L.mzk_set_table_budget(ctypes.c_size_t(0))


def case_merkle():
    fid = rng.choice((FR, M128))
    n = rng.choice([1, 2, 3, 5, 8, 31, 64, 100, 1000, 4096, 5000])
    x = vec(fid, n)
    check("merkle_commit", mz.merkle_commit_field(fid, x) == orc.merkle_commit_field_ref(fid, x), (fid, n))
    if n >= 2:
        t = mz.MerkleTree(fid, x)
        leaves = orc.field_leaves(fid, x)
        i = rng.randrange(n)
        try:
            got = [bytes(b) for b in t.open(i)]
        except mz.MzkError as e:          # Merkle::open does not terminate on the one-leaf half of a three-leaf slice (merkle.rs:36-45): MZK_E_LENGTH
            got = e.code
        ref = orc.merkle_open_ref(i, leaves)
        want = -5 if ref is None else [bytes(b) for b in ref]
        check("merkle_open", got == want, (fid, n, i))
        t.close()


def case_poly():
    fid = rng.choice((FR, M128))
    la, lb = rng.choice([1, 2, 7, 8, 9, 100, 600]), rng.choice([1, 3, 8, 50, 300])
    a, b = vec(fid, la), vec(fid, lb)
    lg = max(1, (la + lb - 2).bit_length())
    w = orc.root_of(fid, lg)
    rc, want = orc.fft_multiply_ref(fid, a, b, w)
    check("fft_multiply", rc == 0 and np.array_equal(mz.fft_multiply(fid, a, b, w), want), (fid, la, lb))
    n = rng.choice([1, 2, 7, 8, 9, 33, 200])
    dom = orc.to_limbs(rng.sample(range(1, 1 << 60), n), NL[fid])
    if rng.random() < 0.4:        # the first n points of a power-of-two subgroup (a STARK's trace domain): the library's inverse-transform path
        lgs = rng.choice([1, 2, 3, 5, 6, 8])
        n = max(1, (1 << lgs) - rng.choice([0, 0, 1, 2, 3, 5, 30, 64, 65]))
        g, pts, acc = orc.root_of(fid, lgs), [], 1
        for _ in range(n):
            pts.append(acc); acc = acc * g % orc.MOD[fid]
        if rng.random() < 0.15 and n > 2:
            pts[rng.randrange(n)] = rng.randrange(1, 1 << 60)          # ... and one that only looks like it
        dom = orc.to_limbs(pts, NL[fid])
    lgr = max(4, (2 * n).bit_length() + 1)
    root, order = orc.root_of(fid, lgr), 1 << lgr
    vals = vec(fid, n)
    rc, want = orc.fast_zerofier_ref(fid, dom, root, order)
    check("fast_zerofier", rc == 0 and np.array_equal(mz.fast_zerofier(fid, dom, root, order), want), (fid, n))
    rc, want = orc.fast_interpolate_ref(fid, dom, vals, root, order)
    check("fast_interpolate", rc == 0 and np.array_equal(mz.fast_interpolate(fid, dom, vals, root, order), want), (fid, n))
    import torch
    batch = rng.choice([1, 2, 5])
    rows = np.stack([vals] + [vec(fid, n) if rng.random() < 0.7 else np.zeros_like(vals) for _ in range(batch - 1)])
    d_v = torch.from_numpy(rows.view(np.int64).reshape(-1).copy()).cuda()
    d_o = torch.full((batch * n * NL[fid],), -1, dtype=torch.int64, device="cuda")
    lens = mz.fast_interpolate_batch_dev(fid, dom, d_v.data_ptr(), batch, root, order, d_o.data_ptr(), torch.cuda.current_stream().cuda_stream)
    got = d_o.cpu().numpy().view(np.uint64).reshape(batch, n, NL[fid])
    ok = rc == 0 and lens[0] == want.shape[0] and np.array_equal(got[0, :lens[0]], want)
    for k in range(batch):
        rk, wk = orc.fast_interpolate_ref(fid, dom, np.ascontiguousarray(rows[k]), root, order)
        ok = ok and rk == 0 and lens[k] == wk.shape[0] and np.array_equal(got[k, :lens[k]], wk) and not got[k, lens[k]:].any()
    check("fast_interpolate_batch_dev", ok, (fid, n, batch))
    cf = vec(fid, rng.choice([1, n, 2 * n + 1]))
    rc, want = orc.fast_evaluate_ref(fid, cf, dom, root, order)
    check("fast_evaluate", rc == 0 and np.array_equal(mz.fast_evaluate(fid, cf, dom, root, order), want), (fid, n))


def case_kzg():
    n = rng.choice([1, 2, 5, 64, 500, 3000])
    alpha = rng.randrange(1, orc.P_FR)
    srs = mz.kzg_setup_g1(alpha, n - 1)
    f = vec(FR, n)
    u = rng.randrange(orc.P_FR)
    y, wpt = mz.kzg_open(f, u, srs)
    fa, fu = orc.poly_eval(FR, f, alpha), orc.poly_eval(FR, f, u)
    q = (fa - fu) * pow((alpha - u) % orc.P_FR, -1, orc.P_FR) % orc.P_FR if alpha != u else None
    ok = y == fu and (q is None or wpt == orc.ec_mul(0, (1, 2), q)) and mz.kzg_commit(f, srs) == orc.ec_mul(0, (1, 2), fa)
    check("kzg_setup_commit_open", ok, n)


def case_fri_fold():
    fid = rng.choice((FR, M128))
    p = orc.MOD[fid]
    lg = rng.choice([0, 1, 2, 3, 6, 10, 13])
    n = 1 << lg
    cw = vec(fid, n)
    alpha, off, om = rng.randrange(p), rng.randrange(1, p), orc.root_of(fid, lg) if lg else 1
    check("fri_fold", np.array_equal(mz.fri_fold(fid, cw, alpha, off, om), orc.fri_fold_ref(fid, cw, alpha, off, om)), (fid, lg))


def case_kzg_next():
    n = rng.choice([2, 5, 33, 64])
    alpha = rng.randrange(1, orc.P_FR)
    srs = mz.kzg_setup_g1(alpha, n - 1)
    f = vec(FR, n)
    us = [rng.randrange(orc.P_FR) for _ in range(rng.choice([1, 2, 3, min(n - 1, 5)]))]
    ys_o, w_o = orc.kzg_batch_open_ref(f, us, srs)
    ys_g, w_g = mz.kzg_batch_open(f, us, srs)
    check("kzg_batch_open", list(ys_g) == list(ys_o) and w_g == w_o, (n, len(us)))
    d = rng.randrange(0, n + 3)
    rc, want = orc.kzg_degree_bound_ref(f, srs, d)
    try:
        got = (0, mz.kzg_prove_degree_bound(f, srs, d))
    except mz.MzkError as e:
        got = (e.code, None)
    check("kzg_prove_degree_bound", (rc == 0 and got == (0, want)) or (rc != 0 and got[0] != 0), (n, d, rc, got[0]))


def case_g2():
    n = rng.choice([0, 1, 2, 7, 40])
    R = orc.P_FR
    G2 = orc.G2_GEN
    base = [orc.g2_mul(G2, rng.randrange(1, R)) for _ in range(min(n, 6))]
    pts = [base[rng.randrange(len(base))] for _ in range(n)] if n else []
    if n > 2 and rng.random() < 0.5:
        pts[1] = orc.G2_INF
    ks = [rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(n)]
    sarr, parr = orc.to_limbs(ks, 4), orc.g2_to_arr(pts)
    check("msm_g2", mz.msm_g2(sarr, parr) == orc.g2_msm_ref(sarr, parr), n)


def case_coset_divide():
    fid = rng.choice((FR, M128))
    p, nl = orc.MOD[fid], NL[fid]
    ll, lr = rng.choice([3, 9, 50, 200]), rng.choice([1, 2, 8, 40])
    lhs, rhs = vec(fid, ll), vec(fid, lr)
    lg = max(3, (ll + 1).bit_length() + rng.choice([0, 1]))
    root, order = orc.root_of(fid, lg), 1 << lg
    offset = rng.randrange(2, 1000)
    rc, want = orc.fast_coset_divide_ref(fid, lhs, rhs, offset, root, order)
    try:
        got = (0, mz.fast_coset_divide(fid, lhs, rhs, offset, root, order))
    except mz.MzkError as e:
        got = (e.code, None)
    check("fast_coset_divide", (rc == 0 and got[0] == 0 and np.array_equal(got[1], want)) or (rc != 0 and got[0] != 0), (fid, ll, lr, lg, rc, got[0]))


CASES = [case_many, case_fri_fold, case_kzg_next, case_g2, case_coset_divide, case_ntt, case_ntt_batch, case_lde, case_scale_columns, case_msm, case_merkle, case_poly, case_kzg]


def run(budget, seed, max_cases=None):
    """Draw cases for `budget` seconds (or max_cases, whichever comes first) from a generator seeded with `seed`.
    Returns (counts per entry point, list of differences)."""
    rng.seed(seed)
    counts.clear()
    del failures[:]
    mz.init(0)
    t0, done = time.time(), 0
    while time.time() - t0 < budget and (max_cases is None or done < max_cases):
        rng.choice(CASES)()
        done += 1
    return dict(counts), list(failures)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    cnt, fails = run(budget, seed)
    print("seed %d, %.0f s: %s; %d difference(s)" % (seed, time.time() - t0, ", ".join("%s %d" % kv for kv in sorted(cnt.items())), len(fails)), flush=True)
    sys.exit(1 if fails else 0)
