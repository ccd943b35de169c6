"""The device arithmetic headers (myzkp_amd/csrc/mzk_field.h, mzk_ec.h), compiled for the host with
bounds assertions, against the oracle.  This checks the exact limb code the HIP kernels execute
(29-bit-limb Montgomery ops, XYZZ group law incl. every exceptional case) without a GPU.  CPU only."""
import ctypes, os, random, subprocess
import numpy as np
import pytest
import orc
from orc import FR, FQ, M128, P_FR, P_FQ, P_M128

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hc():
    src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
    so = os.path.join(HERE, "hostcheck", "libhostcheck.so")
    hdrs = [os.path.join(orc.ROOT, "myzkp_amd", "csrc", h) for h in ("mzk_field.h", "mzk_ec.h", "mzk_g2.h", "mzk_glv.h", "mzk_constants.h")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-DMZK_CHECK_BOUNDS", "-fPIC", "-shared", "-o", so, src])
    return ctypes.CDLL(so)


def w32(v, nw):
    return np.array([(int(v) >> (32 * i)) & 0xFFFFFFFF for i in range(nw)], dtype=np.uint32)


def from_w32(a):
    return sum(int(x) << (32 * i) for i, x in enumerate(a))


def fop(hc, fid, op, a, b):
    nw = 4 if fid == M128 else 8
    out = np.zeros(nw, dtype=np.uint32)
    aa, bb = w32(a, nw), w32(b, nw)
    assert hc.hc_field_op(fid, op, orc.ptr(aa), orc.ptr(bb), orc.ptr(out)) == 0
    return from_w32(out)


def test_field_ops_against_python(hc):
    rng = random.Random(7)
    for fid, p in ((FR, P_FR), (FQ, P_FQ), (M128, P_M128)):
        edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << (p.bit_length() - 1)) - 1]
        vals = edge + [rng.randrange(p) for _ in range(150)]
        for i, a in enumerate(vals):
            b = vals[(i * 7 + 3) % len(vals)]
            assert fop(hc, fid, 0, a, b) == a * b % p
            assert fop(hc, fid, 1, a, b) == a * a % p
            assert fop(hc, fid, 2, a, b) == (a + b) % p
            assert fop(hc, fid, 3, a, b) == (a - b) % p
            assert fop(hc, fid, 5, a, b) == (-a) % p
            assert fop(hc, fid, 6, a, b) == pow(2 * (a + b) - b, 2, p)
            assert fop(hc, fid, 7, a, b) == a * b % p
            assert fop(hc, fid, 10, a, b) == (pow(a, -1, p) if a else 0)             # safegcd (divsteps), every value
            assert fop(hc, fid, 11, a, b) == (pow(2 * a, -1, p) if a else 0)         # ... of a lazily reduced input
            if i < 40:
                assert fop(hc, fid, 4, a, b) == (pow(a, -1, p) if a else 0)          # binary extended Euclid
                assert fop(hc, fid, 9, a, b) == (pow(2 * a, -1, p) if a else 0)      # ... of a lazily reduced input
            if i < 12:
                assert fop(hc, fid, 8, a, b) == (pow(a, -1, p) if a else 0)          # Fermat ladder
                assert fop(hc, fid, 0, a, b) == orc.field_op("mul", fid, a, b)


def test_safegcd_inverse_many_values(hc):
    """fe_inv_safegcd (62 divsteps per batch) on structured and random inputs: powers of two and their neighbours
    (long zero runs in g), values next to p, small values, and 1500 random ones per field."""
    rng = random.Random(99)
    for fid, p in ((FQ, P_FQ), (FR, P_FR), (M128, P_M128)):
        vals = [1, 2, 3, p - 1, p - 2, p - 3, (p + 1) // 2, (p - 1) // 2]
        for k in range(1, p.bit_length()):
            vals += [(1 << k) % p, ((1 << k) - 1) % p, ((1 << k) + 1) % p, (p - (1 << k)) % p]
        vals += [rng.randrange(1, p) for _ in range(1500)]
        vals += [rng.randrange(1, 1 << 64) for _ in range(100)]
        for a in vals:
            if a:
                assert fop(hc, fid, 10, a, 0) == pow(a, -1, p), (fid, a)


def test_pack_roundtrip(hc):
    rng = random.Random(8)
    for fid, nw, bits in ((FR, 8, 256), (M128, 4, 128)):
        for _ in range(50):
            v = rng.getrandbits(bits)
            a, out = w32(v, nw), np.zeros(nw, dtype=np.uint32)
            hc.hc_pack_roundtrip(fid, orc.ptr(a), orc.ptr(out))
            assert from_w32(out) == v


def ptw(P):
    return np.concatenate([w32(P[0], 8), w32(P[1], 8)])


def g1op(hc, op, P, Q=(0, 0)):
    out = np.zeros(16, dtype=np.uint32)
    p, q = ptw(P), ptw(Q)
    assert hc.hc_g1_op(op, orc.ptr(p), orc.ptr(q), orc.ptr(out)) == 0
    return (from_w32(out[:8]), from_w32(out[8:]))


def test_group_law_and_exceptions(hc):
    pts = orc.arr_to_pts(orc.synth_points(11, 12))
    INF = (0, 0)
    neg = lambda P: (P[0], (P_FQ - P[1]) % P_FQ)
    cases = [(pts[0], pts[1]), (pts[2], pts[2]), (pts[3], neg(pts[3])), (INF, pts[4]), (pts[5], INF), (INF, INF),
             (pts[6], pts[7]), ((1, 2), (1, 2)), ((1, 2), (1, P_FQ - 2))]
    for P, Q in cases:
        want = orc.ec_add(0, P, Q)
        assert g1op(hc, 0, P, Q) == want, ("madd", P, Q)
        assert g1op(hc, 1, P, Q) == want, ("add", P, Q)
        assert g1op(hc, 5, P, Q) == want, ("signed madd, +", P, Q)
        assert g1op(hc, 4, P, neg(Q) if Q != INF else Q) == want, ("signed madd, -", P, Q)      # P - (-Q) through the sign path
    for P in pts[:6] + [(1, 2), INF]:
        want = orc.ec_mul(0, P, 2)
        assert g1op(hc, 2, P) == want
        assert g1op(hc, 3, P) == want


def test_scalar_mul_through_xyzz(hc):
    rng = random.Random(9)
    ks = [0, 1, 2, P_FR - 1, P_FR, rng.randrange(P_FR), rng.randrange(P_FR)]
    P = orc.arr_to_pts(orc.synth_points(3, 1))[0]
    for k in ks:
        out = np.zeros(16, dtype=np.uint32)
        p, kk = ptw(P), w32(k, 8)
        hc.hc_g1_mul(orc.ptr(p), orc.ptr(kk), orc.ptr(out))
        assert (from_w32(out[:8]), from_w32(out[8:])) == orc.ec_mul(0, P, k)


def test_running_sum_through_packed_storage(hc):
    pts = orc.arr_to_pts(orc.synth_points(21, 40))
    seq = pts[:10] + [pts[3], pts[3], (0, 0), (pts[5][0], P_FQ - pts[5][1])] + pts[10:]
    arr = np.concatenate([ptw(P) for P in seq])
    out = np.zeros(16, dtype=np.uint32)
    hc.hc_g1_sum(orc.ptr(arr), len(seq), orc.ptr(out))
    want = (0, 0)
    for P in seq:
        want = orc.ec_add(0, want, P)
    assert (from_w32(out[:8]), from_w32(out[8:])) == want


def _g2op(hc, op, A, B=None, k=0):
    a = orc.g2_to_arr([A]).view(np.uint32).reshape(-1).copy()
    b = orc.g2_to_arr([B if B is not None else orc.G2_INF]).view(np.uint32).reshape(-1).copy()
    kw = w32(k, 8)
    out = np.zeros(32, dtype=np.uint32)
    rc = hc.hc_g2_op(op, orc.ptr(a), orc.ptr(b), orc.ptr(kw), orc.ptr(out))
    assert rc == 0, rc
    return orc.arr_to_g2(out.view(np.uint64))[0]


def test_g2_fq2_arithmetic_and_group_law_with_bounds(hc):
    """mzk_g2.h under -DMZK_CHECK_BOUNDS: Fq2 mul / inverse, XYZZ add (mixed and general), double, scalar mul over Fq2
    against the oracle (which test_oracle_g2.py pins to the reference's test_g2)."""
    rng = random.Random(21)
    G = orc.G2_GEN
    q = P_FQ
    for _ in range(20):
        x = (rng.randrange(q), rng.randrange(q)); y = (rng.randrange(q), rng.randrange(q))
        got = _g2op(hc, 3, (x, (0, 0)), (y, (0, 0)))[0]
        assert got == ((x[0] * y[0] - x[1] * y[1]) % q, (x[0] * y[1] + x[1] * y[0]) % q)
        inv = _g2op(hc, 4, (x, (0, 0)))[0]
        n = pow(x[0] * x[0] + x[1] * x[1], -1, q)
        assert inv == (x[0] * n % q, -x[1] * n % q)
    pts = [orc.g2_mul(G, rng.randrange(1, P_FR)) for _ in range(6)]
    neg = lambda P: (P[0], ((-P[1][0]) % q, (-P[1][1]) % q))
    for P, Q in [(pts[0], pts[1]), (pts[2], pts[2]), (pts[3], neg(pts[3])), (orc.G2_INF, pts[4]), (pts[5], orc.G2_INF),
                 (orc.G2_INF, orc.G2_INF)]:
        assert _g2op(hc, 0, P, Q) == orc.g2_add(P, Q)
    for P in pts[:3] + [orc.G2_INF]:
        assert _g2op(hc, 1, P) == orc.g2_add(P, P)
    for k in [0, 1, 2, 3, P_FR - 1, P_FR, rng.randrange(P_FR), (1 << 200) + 5]:
        assert _g2op(hc, 2, pts[0], k=k) == orc.g2_mul(pts[0], k), k


def test_glv_split_identity_and_bounds(hc):
    """mzk_glv.h: k1 + k2 lambda == k (mod r), both parts below 2^126 (asserted inside the header too), and the
    endomorphism itself: lambda P == (beta x, y) on curve points"""
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
    rng = random.Random(77)
    ks = [0, 1, 2, P_FR - 1, P_FR - 2, lam, lam + 1, P_FR - lam, (P_FR - 1) // 2, 1 << 127, (1 << 128) - 1, 1 << 253] + \
         [rng.randrange(P_FR) for _ in range(3000)] + [rng.getrandbits(b) for b in range(1, 254, 3)]
    worst = 0
    for k in ks:
        out = np.zeros(10, dtype=np.uint32)
        assert hc.hc_glv_split(orc.ptr(w32(k, 8)), orc.ptr(out)) == 0
        m1, n1, m2, n2 = from_w32(out[0:4]), int(out[4]), from_w32(out[5:9]), int(out[9])
        k1, k2 = (-m1 if n1 else m1), (-m2 if n2 else m2)
        assert (k1 + k2 * lam - k) % P_FR == 0, k
        assert m1 < 1 << 126 and m2 < 1 << 126
        worst = max(worst, m1, m2)
    assert worst > 1 << 120                       # the parts really are half-length, not trivially small
    for kk in (1, 5, 12345678901234567890):
        P = orc.ec_mul(0, (1, 2), kk)
        assert orc.ec_mul(0, P, lam) == (beta * P[0] % P_FQ, P[1])


def test_ntt_stage_schedule_with_lazy_first_stages_holds_its_bounds(hc):
    """The kernels' in-tile stage schedule (mzk_ntt.hip tile_stages: stage pairs, first stage without carry propagation)
    restated on a plain array in hostcheck.cpp and run with the bounds assertions on: every level count 1..10, seeded random
    inputs, all p - 1, and raw all-ones words (non-canonical input must not break a limb bound either); canonical inputs
    against the oracle's transform (ntt.rs:7-48)."""
    for fid, nl in ((FR, 4), (M128, 2)):
        p = orc.MOD[fid]
        nw = 2 * nl
        for lgn in range(1, 11):
            n = 1 << lgn
            w = orc.root_of(fid, lgn)
            tw = orc.to_limbs([pow(w, j, p) for j in range(max(n // 2, 1))], nl).view(np.uint32).reshape(-1)
            cases = [orc.synth_vector(fid, 40 + lgn, n, 1), orc.to_limbs([p - 1] * n, nl), orc.to_limbs([1] + [0] * (n - 1), nl),
                     orc.to_limbs([p - 2] * n, nl)]          # p - 2: every 29-bit limb of M128 at its maximum (the signed lazy sums' worst case)
            for x in cases:
                out = np.zeros(n * nw, dtype=np.uint32)
                assert hc.hc_ntt_tile(fid, orc.ptr(np.ascontiguousarray(x).view(np.uint32).reshape(-1)), lgn, orc.ptr(tw), orc.ptr(out)) == 0
                rc, want = orc.ntt_fast(fid, w, np.ascontiguousarray(x), False, 1)
                assert rc == 0 and np.array_equal(out.view(np.uint64).reshape(n, nl), want), (fid, lgn)
            raw = np.full(n * nw, 0xFFFFFFFF, dtype=np.uint32)          # bounds only: the asserts inside must hold
            out = np.zeros(n * nw, dtype=np.uint32)
            assert hc.hc_ntt_tile(fid, orc.ptr(raw), lgn, orc.ptr(tw), orc.ptr(out)) == 0


def _limbs29(v, n=9):
    return [(v >> (29 * i)) & 0x1fffffff for i in range(n)]


def test_shoup_product_matches_its_integer_model_and_bounds(hc):
    """fe_shoup_mul (mzk_field.h; the asm form of the NTT's wave-uniform twiddles computes the same columns): x * w mod p
    for constant w with wq = floor(w 2^261 / p) -- result congruent, below 4 p, limbs normalised, for normalised and for the
    widest lazy x (limbs up to 3 * 2^30), with the column-overflow assertion of the host build on."""
    import random
    rng = random.Random(21)
    p = P_FR
    beta = 1 << 261
    for case in range(3000):
        wv = rng.randrange(p) if case % 6 else [0, 1, p - 1, 2, p // 2, (1 << 253)][(case // 6) % 6]
        lim = [0x1fffffff, 3 << 30, (1 << 30) + (1 << 29)][case % 3]
        x = [rng.randrange(lim + 1) for _ in range(9)]
        if case % 40 == 0:
            x = [lim] * 9
        x[8] = rng.randrange(1 << 27)
        xv = sum(v << (29 * i) for i, v in enumerate(x))
        out = np.zeros(9, dtype=np.uint32)
        assert hc.hc_shoup_mul(orc.ptr(np.array(x, dtype=np.uint32)), orc.ptr(np.array(_limbs29(wv), dtype=np.uint32)),
                               orc.ptr(np.array(_limbs29(wv * beta // p), dtype=np.uint32)), orc.ptr(out)) == 0
        rv = sum(int(v) << (29 * i) for i, v in enumerate(out))
        assert rv % p == xv * wv % p and rv < 4 * p and all(int(v) <= 0x1fffffff for v in out), case


def test_ntt_stage_schedule_with_shoup_products_holds_its_bounds(hc):
    """The stage schedule again with the precomputed-quotient products in the stage pairs whose twiddles a whole wave shares
    (tile of 2^lgn rows x 2^lgc columns as the kernels cut it: 2^10 x 4 and 2^8 x 4), bounds assertions on, results against
    the oracle's transform."""
    p = P_FR
    for lgn, lgc in ((10, 2), (8, 2), (7, 3), (6, 4), (4, 6)):
        n = 1 << lgn
        w = orc.root_of(FR, lgn)
        pw = [pow(w, j, p) for j in range(n // 2)]
        tw = orc.to_limbs(pw, 4).view(np.uint32).reshape(-1)
        sh = np.array([_limbs29(v) + _limbs29(v * (1 << 261) // p) for v in pw], dtype=np.uint32).reshape(-1)
        cases = [orc.synth_vector(FR, 140 + lgn, n, 1), orc.to_limbs([p - 1] * n, 4), orc.to_limbs([1] + [0] * (n - 1), 4)]
        for x in cases:
            out = np.zeros(n * 8, dtype=np.uint32)
            assert hc.hc_ntt_tile_shoup(orc.ptr(np.ascontiguousarray(x).view(np.uint32).reshape(-1)), lgn, lgc, orc.ptr(tw), orc.ptr(sh), orc.ptr(out)) == 0
            rc, want = orc.ntt_fast(FR, w, np.ascontiguousarray(x), False, 1)
            assert rc == 0 and np.array_equal(out.view(np.uint64).reshape(n, 4), want), (lgn, lgc)
        raw = np.full(n * 8, 0xFFFFFFFF, dtype=np.uint32)
        out = np.zeros(n * 8, dtype=np.uint32)
        assert hc.hc_ntt_tile_shoup(orc.ptr(raw), lgn, lgc, orc.ptr(tw), orc.ptr(sh), orc.ptr(out)) == 0


def test_m128_sparse_product_signed_and_unsigned(hc):
    """fe_mul_sparse (mzk_field.h; FeAsm<M128Params>::mul / smul compute the same columns): a b / 2^145 mod p for
    p = 3256 * 2^116 + 1 without a multiplication by -p^-1 -- congruent, inside [T/R + c p, T/R + (1 + c) p + p / 2^29 + 3), low limbs
    normalised, for unsigned lazy and for SIGNED lazy operands (i32 limbs up to +-(2^31 - 1)), a non-negative operand giving a
    non-negative result; the column bound assertions of the host build are on."""
    import random
    rng = random.Random(33)
    p, R = P_M128, 1 << 145
    val = lambda l: sum(int(v) << (29 * i) for i, v in enumerate(l))
    for case in range(6000):
        signed = case % 2
        kind = (case // 2) % 5
        if kind == 0:
            a = [rng.randrange(-(1 << 31) + 1, 1 << 31) if signed else rng.randrange(1 << 32) for _ in range(5)]
        elif kind == 1:
            a = _limbs29(rng.randrange(p), 5)
        elif kind == 2:
            a = [(1 << 31) - 1] * 5 if signed else [(1 << 32) - 1] * 5
        elif kind == 3:
            a = [-((1 << 31) - 1)] * 5 if signed else [0] * 5
        else:
            a = [0, 0, 0, 0, 0]
            a[rng.randrange(5)] = rng.randrange(1 << 29)
        if abs(val(a)) >= 1 << 144:
            a[4] = rng.randrange(-(1 << 26), 1 << 26) if signed else rng.randrange(1 << 26)
        b = _limbs29(rng.randrange(p), 5) if case % 7 else [0x1fffffff] * 4 + [3256]
        for c in (0, 1):
            out = np.zeros(5, dtype=np.uint32)
            aa = np.array([v & 0xFFFFFFFF for v in a], dtype=np.uint32)
            assert hc.hc_m128_mul_sparse(signed, c, orc.ptr(aa), orc.ptr(np.array(b, dtype=np.uint32)), orc.ptr(out)) == 0
            r = [int(v) for v in out]
            if signed and r[4] >= 1 << 31:
                r[4] -= 1 << 32
            A, B, V = val(a), val(b), val(r)
            assert (V * R - A * B) % p == 0, case
            assert all(0 <= v <= 0x1fffffff for v in r[:4])
            assert A * B // R + c * p <= V <= A * B // R + (1 + c) * p + (p >> 29) + 3, case
            if A >= 0:
                assert V >= 0


def test_m128_signed_lazy_reduction_is_canonical(hc):
    """fe_sreduce: any signed lazy value below 2^12 p in magnitude -> [0, p); the borrow path (limb 0 below the folded quotient)
    is forced as well: multiples of 2^116 * 3256, of p, values just around them."""
    import random
    rng = random.Random(34)
    p = P_M128
    lim = (1 << 12) * p

    def lazy_limbs(x, spread):
        # a signed-limb representation of x with limbs pushed away from the carried form (floor semantics: x may be negative)
        l = [(x >> (29 * i)) & 0x1fffffff for i in range(4)] + [x >> 116]
        for i in range(4):
            k = rng.randrange(-spread, spread + 1)
            l[i] -= k << 29
            l[i + 1] += k
        return l

    xs = [rng.randrange(-lim + 1, lim) for _ in range(400)]
    for q in list(range(0, 40)) + [4094, 4095]:
        for d in (-3, -1, 0, 1, 2, 5):
            xs += [q * p + d, -q * p + d, ((q * 3256) << 116) + d]          # the last: top limb a multiple of 3256, limb 0 < q
    top = 3 * ((1 << 29) - 1)            # the widest limb a stage pair leaves
    reps = {sum(v << (29 * i) for i, v in enumerate(l)): l for l in ([top] * 4 + [4 * 4095], [top, 0, 0, 0, 0], [-top] * 4 + [-40000],
                                                                        [top, top, -top, top, 77])}
    for x in xs + list(reps):
        if abs(x) >= lim:
            continue
        for spread in (0, 1):
            l = reps[x] if x in reps else lazy_limbs(x, spread)
            assert sum(v << (29 * i) for i, v in enumerate(l)) == x
            assert all(abs(v) <= 3 * ((1 << 29) - 1) for v in l)
            out = np.zeros(4, dtype=np.uint32)
            aa = np.array([v & 0xFFFFFFFF for v in l], dtype=np.uint32)
            assert hc.hc_m128_sreduce(orc.ptr(aa), orc.ptr(out)) == 0
            assert from_w32(out) == x % p, (x, spread)
