"""GPU parity: NTT / iNTT / coset LDE / polynomial products through the C ABI vs the oracle and the
committed golden vectors.  Bit-exact (integer arithmetic)."""
import numpy as np
import pytest
import orc
from orc import FR, M128, I

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


def _fid(name):
    return {"Fr": FR, "M128": M128}[name]


def test_golden_vectors(mz):
    g = orc.golden("ntt_vectors.json")
    for c in g["cases"]:
        f = _fid(c["field"])
        nl = orc.LIMBS[f]
        k = c["kind"]
        if k in ("ntt", "intt"):
            out = mz.ntt(f, int(c["root"]), orc.to_limbs(I(c["input"]), nl), inverse=(k == "intt"))
        elif k == "coset":
            out = mz.coset_lde(f, orc.to_limbs(I(c["input"]), nl), int(c["offset"]), int(c["generator"]), c["order"])
        elif k == "fft_multiply":
            out = mz.fft_multiply(f, orc.to_limbs(I(c["a"]), nl), orc.to_limbs(I(c["b"]), nl), int(c["omega"]))
        else:
            out = mz.fast_multiply(f, orc.to_limbs(I(c["a"]), nl), orc.to_limbs(I(c["b"]), nl), int(c["root"]), c["root_order"])
        assert orc.from_limbs(out) == I(c["output"]) if len(c["output"]) else out.shape[0] == 0, (k, c["field"])


def test_reference_test_ntt(mz):
    # ntt.rs:346-374
    n, w = 256, orc.m128_root(8)
    coef = orc.to_limbs(list(range(1, n + 1)), 2)
    vals = mz.ntt(M128, w, coef)
    v = orc.from_limbs(vals)
    for i in range(0, n, 17):
        assert v[i] == orc.poly_eval(M128, coef, orc.field_pow(M128, w, i))
    assert orc.from_limbs(mz.intt(M128, w, vals)) == list(range(1, n + 1))


@pytest.mark.parametrize("fid", [FR, M128])
@pytest.mark.parametrize("lg", list(range(0, 15)))
def test_all_small_sizes_vs_ref(mz, fid, lg):
    n = 1 << lg
    v = orc.synth_vector(fid, 100 + lg, n)
    w = orc.root_of(fid, lg) if lg else 1
    if lg <= 11:
        rc, want = orc.ntt_ref(fid, w, v)
    else:
        rc, want = orc.ntt_fast(fid, w, v)
    assert rc == 0
    got = mz.ntt(fid, w, v)
    assert np.array_equal(got, want)
    rc, wanti = orc.ntt_fast(fid, w, v, inverse=True) if lg else (0, v)
    assert np.array_equal(mz.intt(fid, w, v), wanti)


@pytest.mark.parametrize("fid,lg", [(FR, 16), (M128, 17), (FR, 20), (M128, 20), (FR, 22)])
def test_large_sizes_vs_fast_oracle(mz, fid, lg):
    n = 1 << lg
    v = orc.synth_vector(fid, 7 + lg, n)
    w = orc.root_of(fid, lg)
    rc, want = orc.ntt_fast(fid, w, v, threads=8)
    assert rc == 0
    got = mz.ntt(fid, w, v)
    assert np.array_equal(got, want)
    back = mz.intt(fid, w, got)
    assert np.array_equal(back, v)


def test_noncanonical_edge_values(mz):
    # all p-1 / zeros / ones inputs; p - 2 and 2^116 - 1: every 29-bit limb of an M128 element at its maximum -- the worst case of the
    # signed lazy butterflies (limb sums up to 4 (2^29 - 1)) and, through X[0] = n x, of the borrow path of the final reduction --
    # in every tile geometry (one pass, two small-tile passes, the 2^10-level tiles of 2^20)
    for fid in (FR, M128):
        p = orc.MOD[fid]
        for lg in (4, 9, 12, 15, 20):
            for fill in (0, 1, p - 1, p - 2, (1 << 116) - 1, p - 64):
                if lg != 12 and fill in (0, 1):
                    continue
                n = 1 << lg
                v = orc.to_limbs([fill] * n, orc.LIMBS[fid])
                w = orc.root_of(fid, lg)
                rc, want = orc.ntt_fast(fid, w, v, threads=8)
                got = mz.ntt(fid, w, v)
                assert rc == 0 and np.array_equal(got, want), (fid, lg, fill)
                assert np.array_equal(mz.intt(fid, w, got), v), (fid, lg, fill)


def test_error_behaviour(mz):
    v = orc.to_limbs([1, 2, 3, 4, 5, 6, 7, 8], 2)
    with pytest.raises(mz.MzkError) as e:      # ntt.rs:8-11
        mz.ntt(M128, orc.m128_root(3), v[:6])
    assert e.value.code == -2
    with pytest.raises(mz.MzkError) as e:      # ntt.rs:15-18
        mz.ntt(M128, orc.m128_root(4), v)
    assert e.value.code == -3
    with pytest.raises(mz.MzkError) as e:      # ntt.rs:19-22
        mz.ntt(M128, orc.m128_root(2), v)
    assert e.value.code == -4
    with pytest.raises(mz.MzkError) as e:      # ntt.rs:265 usize underflow
        mz.coset_lde(M128, v, 3, orc.m128_root(2), 4)
    assert e.value.code == -5
    assert mz.ntt(M128, 1, v[:0]).shape[0] == 0
    assert np.array_equal(mz.ntt(M128, 12345, v[:1]), v[:1])      # len 1 returned unchanged, root unchecked


def test_coset_lde_stark_parameters(mz):
    # initialize_fast_stark_m128 (fast_stark.rs:573-616): offset = generator of order 2^119
    n_coef, order = 1 << 10, 1 << 14
    c = orc.synth_vector(M128, 5, n_coef)
    g = orc.m128_root(14)
    out = mz.coset_lde(M128, c, orc.M128_GEN, g, order)
    # oracle: scale then fast NTT
    scaled = [x * pow(orc.M128_GEN, i, orc.P_M128) % orc.P_M128 for i, x in enumerate(orc.from_limbs(c))]
    rc, want = orc.ntt_fast(M128, g, orc.to_limbs(scaled + [0] * (order - n_coef), 2))
    assert rc == 0 and np.array_equal(out, want)
    rc, want_ref = orc.coset_ref(M128, c[:64], orc.M128_GEN, orc.m128_root(8), 256)
    assert np.array_equal(mz.coset_lde(M128, c[:64], orc.M128_GEN, orc.m128_root(8), 256), want_ref)


def test_linearity_at_full_size(mz):
    # size-independent property at BASELINE size 2^20: NTT(a + b) == NTT(a) + NTT(b)
    lg, fid = 20, FR
    n = 1 << lg
    p = orc.P_FR
    a = orc.synth_vector(fid, 1, n)
    b = orc.synth_vector(fid, 2, n)
    w = orc.root_of(fid, lg)
    A, B = mz.ntt(fid, w, a), mz.ntt(fid, w, b)
    idx = np.random.RandomState(0).randint(0, n, 64)
    ai, bi = orc.from_limbs(a), orc.from_limbs(b)
    s = orc.to_limbs([(x + y) % p for x, y in zip(ai, bi)], 4)
    Sv = mz.ntt(fid, w, s)
    Al, Bl, Sl = orc.from_limbs(A[idx]), orc.from_limbs(B[idx]), orc.from_limbs(Sv[idx])
    assert all((x + y) % p == z for x, y, z in zip(Al, Bl, Sl))


@pytest.mark.parametrize("fid", [M128, FR])
def test_coset_lde_fused_prescale_and_table_cache(mz, fid):
    """multi-pass sizes fuse Polynomial::scale + padding into the first pass and keep the offset-power tables between
    calls: alternate offsets, sizes and coefficient counts so that hits, misses and regrown tables all occur"""
    p = orc.MOD[fid]
    offs = [orc.M128_GEN % p, 5, 7, 5]
    for lg, ncoef in ((12, 1 << 10), (12, 1000), (13, 1 << 13), (12, 1 << 10), (11, 3), (14, 1 << 12)):
        order = 1 << lg
        gen = orc.root_of(fid, lg)
        coef = orc.synth_vector(fid, 70 + lg + ncoef, ncoef)
        for off in offs:
            got = mz.coset_lde(fid, coef, off, gen, order)
            want = orc.coset_ref(fid, coef, off, gen, order)[1] if lg <= 12 else None
            if want is None:      # literal oracle is O(n log^2 n): above 2^12 use the fast oracle on the scaled, padded vector
                vals = orc.from_limbs(coef)
                acc, sc = 1, []
                for x in vals:
                    sc.append(x * acc % p); acc = acc * off % p
                arr = np.zeros((order, orc.LIMBS[fid]), dtype=np.uint64)
                arr[:ncoef] = orc.to_limbs(sc, orc.LIMBS[fid])
                rc, want = orc.ntt_fast(fid, gen, arr)
                assert rc == 0
            assert np.array_equal(got, want), (lg, ncoef, off)


@pytest.mark.parametrize("fid", [0, 1])
def test_batched_transforms_equal_single_calls(fid):
    """mzk_ntt_batch: `batch` transforms stored back to back, one launch per pass -- every row bit-identical to the single
    call (ntt.rs:7-64), forward and inverse, sizes on every code path (one level, two levels, the 2^20 large tiles are
    covered by the single-transform tests), batch counts that do not fill the last workgroup, and the trivial sizes."""
    import myzkp_amd as mz
    mz.init(0)
    rng = np.random.default_rng(77 + fid)
    nl = mz.LIMBS[fid]
    for lg, batch in ((0, 3), (1, 5), (3, 7), (6, 33), (10, 9), (12, 5), (14, 3)):
        n = 1 << lg
        root = mz.root_of_unity(fid, lg)
        cols = np.stack([orc.synth_vector(fid, 9000 + 31 * k + lg, n) for k in range(batch)])
        got = mz.ntt_batch(fid, root, cols)
        back = mz.ntt_batch(fid, root, got, inverse=True)
        for k in range(batch):
            assert np.array_equal(got[k], mz.ntt(fid, root, cols[k])), (lg, k)
        assert np.array_equal(back.reshape(cols.shape), cols)
    assert mz.ntt_batch(fid, mz.root_of_unity(fid, 4), np.zeros((0, 16, nl), dtype=np.uint64)).shape[0] == 0
    with pytest.raises(mz.MzkError):
        mz.ntt_batch(fid, mz.root_of_unity(fid, 4), np.zeros((2, 12, nl), dtype=np.uint64))      # not a power of two


@pytest.mark.parametrize("fid", [0, 1])
def test_batched_coset_lde_equals_single_calls(fid):
    """mzk_coset_lde_batch: fast_coset_evaluate (ntt.rs:254-269) of several polynomials onto one coset -- every row equal to
    the single call: single-pass orders (the scale / pad kernel runs over the batch), multi-pass orders (scale and padding
    fused into the first pass, coefficient vectors n_coef apart), ragged n_coef, n_coef == order, batch of one and of none."""
    import myzkp_amd as mz
    mz.init(0)
    nl = mz.LIMBS[fid]
    off = orc.M128_GEN if fid == 1 else 5
    for lg_order, n_coef, batch in ((4, 5, 3), (10, 1024, 2), (10, 300, 7), (12, 1000, 5), (14, 4096, 3), (16, 16384 + 7, 2), (12, 0, 2), (12, 9, 1)):
        order = 1 << lg_order
        gen = mz.root_of_unity(fid, lg_order)
        coefs = np.stack([orc.synth_vector(fid, 4000 + 13 * k + lg_order, max(n_coef, 1))[:n_coef] for k in range(batch)]).reshape(batch, n_coef, nl)
        got = mz.coset_lde_batch(fid, coefs, off, gen, order)
        assert got.shape == (batch, order, nl)
        for k in range(batch):
            assert np.array_equal(got[k], mz.coset_lde(fid, coefs[k], off, gen, order)), (lg_order, n_coef, k)
    # 2^20: a batch switches the plan to the small tiles (three passes), the single call uses the large ones (two): both
    # geometries, their own pre-scale tables, same numbers
    order, n_coef, batch = 1 << 20, (1 << 18) + 3, 3
    gen = mz.root_of_unity(fid, 20)
    coefs = np.stack([orc.synth_vector(fid, 5100 + k, n_coef) for k in range(batch)])
    got = mz.coset_lde_batch(fid, coefs, off, gen, order)
    for k in (0, batch - 1):
        assert np.array_equal(got[k], mz.coset_lde(fid, coefs[k], off, gen, order))
    tb = mz.ntt_batch(fid, gen, got)
    assert np.array_equal(tb[1], mz.ntt(fid, gen, got[1]))
    assert mz.coset_lde_batch(fid, np.zeros((0, 4, nl), dtype=np.uint64), off, mz.root_of_unity(fid, 4), 16).shape == (0, 16, nl)
    with pytest.raises(mz.MzkError):
        mz.coset_lde_batch(fid, np.zeros((2, 40, nl), dtype=np.uint64), off, mz.root_of_unity(fid, 5), 32)      # n_coef > order
