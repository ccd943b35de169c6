"""GPU parity for the Merkle commitments of FRI / STARK codewords (merkle.rs:15-46 as called
from fri.rs:160-166 and fri.rs:236-249): roots and authentication paths bit-identical to the oracle."""
import ctypes, hashlib, random
import numpy as np
import pytest
import orc
from orc import FR, M128

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


def _edge_vector(fid, seed, n):
    """uniform elements with short encodings mixed in: 0 (9-byte leaf), small values, exact digit boundaries"""
    nl = orc.LIMBS[fid]
    arr = orc.synth_vector(fid, seed, n)
    specials = [0, 1, 255, (1 << 32) - 1, 1 << 32, (1 << 64) - 1, 1 << 64, 1 << 96, orc.MOD[fid] - 1, (1 << 120) + 5]
    for i, v in enumerate(specials):
        if 3 * i + 1 < n:
            arr[3 * i + 1] = orc.to_limbs([v % orc.MOD[fid]], nl)[0]
    return arr


@pytest.mark.parametrize("fid", [M128, FR])
@pytest.mark.parametrize("lg", [0, 1, 2, 3, 9, 10, 11, 13])
def test_field_tree_root_matches_oracle(mz, fid, lg):
    n = 1 << lg
    arr = _edge_vector(fid, 300 + lg, n)
    leaves = orc.field_leaves(fid, arr)
    want = orc.merkle_commit_ref(leaves)
    assert mz.merkle_commit_field(fid, arr) == want
    t = mz.MerkleTree(fid, arr)
    assert t.root() == want
    t.close()


@pytest.mark.parametrize("fid,lg", [(M128, 1), (M128, 2), (M128, 12), (FR, 5), (FR, 11)])
def test_open_paths_match_oracle_and_verify(mz, fid, lg):
    n = 1 << lg
    arr = _edge_vector(fid, 400 + lg, n)
    leaves = orc.field_leaves(fid, arr)
    root = orc.merkle_commit_ref(leaves)
    t = mz.MerkleTree(fid, arr)
    rnd = random.Random(lg)
    for idx in sorted({0, 1, n - 1, n // 2, 4 % n, 5 % n} | {rnd.randrange(n) for _ in range(6)}):
        path = t.open(idx)
        assert path == orc.merkle_open_ref(idx, leaves), idx
        assert orc.merkle_verify_ref(root, idx, path, leaves[idx])
        assert not orc.merkle_verify_ref(root, idx, path, leaves[idx ^ 1] + b"x")
    t.close()


def test_reference_merkle_test_on_gpu(mz):
    """merkle.rs:76-93: leaf1..leaf4, open index 2"""
    leaves = [b"leaf1", b"leaf2", b"leaf3", b"leaf4"]
    t = mz.MerkleTree(leaves=leaves)
    root = t.root()
    assert root == orc.merkle_commit_ref(leaves)
    proof = t.open(2)
    assert proof == orc.merkle_open_ref(2, leaves)
    assert orc.merkle_verify_ref(root, 2, proof, leaves[2]) and not orc.merkle_verify_ref(root, 2, proof, leaves[3])


@pytest.mark.parametrize("n", [1, 2, 8, 1024, 2048])
def test_byte_leaves_ragged_lengths_and_multi_block(mz, n):
    rnd = random.Random(n)
    leaves = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 9, 25, 41, 67, 68, 69, 135, 136, 137, 300]))) for _ in range(n)]
    t = mz.MerkleTree(leaves=leaves)
    assert t.root() == orc.merkle_commit_ref(leaves)
    if n >= 2:
        for idx in {0, n - 1, n // 3}:
            assert t.open(idx) == orc.merkle_open_ref(idx, leaves)


def test_contract_violations(mz):
    with pytest.raises(mz.MzkError) as e:
        mz.merkle_commit_field(M128, np.zeros((0, 2), dtype=np.uint64))
    assert e.value.code == -5
    bad = orc.synth_vector(M128, 1, 4)
    bad[2] = orc.to_limbs([orc.MOD[M128]], 2)[0]
    with pytest.raises(mz.MzkError) as e:
        mz.merkle_commit_field(M128, bad)
    assert e.value.code == -6
    t = mz.MerkleTree(M128, orc.synth_vector(M128, 1, 4))
    with pytest.raises(mz.MzkError):
        t.open(4)
    one = mz.MerkleTree(M128, orc.synth_vector(M128, 1, 1))
    with pytest.raises(mz.MzkError):
        one.open(0)


@pytest.mark.parametrize("n", [3, 5, 6, 7, 9, 12, 13, 100, 1000, 1025, 4097, 12345])
def test_ragged_leaf_counts_match_the_reference_recursion(mz, n):
    """merkle.rs:15-25 splits at mid = len / 2 for ANY length: byte leaves of mixed lengths, and field codewords, on
    leaf counts that are not powers of two; roots and every terminating authentication path equal the oracle's
    literal recursion, the non-terminating ones (leaf alone in its subtree: Merkle::open recurses forever) are errors."""
    rnd = random.Random(n)
    leaves = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 9, 25, 32, 41, 135, 136, 137, 300]))) for _ in range(n)]
    t = mz.MerkleTree(leaves=leaves)
    root = t.root()
    assert root == orc.merkle_commit_ref(leaves)
    opened = refused = 0
    for idx in (range(n) if n <= 13 else sorted({0, 1, 2, n - 1, n - 2, n // 2, n // 3} | {rnd.randrange(n) for _ in range(10)})):
        want = orc.merkle_open_ref(idx, leaves)
        if want is None:
            with pytest.raises(mz.MzkError) as e:
                t.open(idx)
            assert e.value.code == -5
            refused += 1
        else:
            path = t.open(idx)
            assert path == want, idx
            opened += 1
    assert opened > 0 and (refused > 0 or n > 13)      # every ragged count has a three-leaf slice somewhere
    t.close()
    for fid in (M128, FR):
        arr = _edge_vector(fid, 500 + n, n)
        want = orc.merkle_commit_ref(orc.field_leaves(fid, arr))
        assert mz.merkle_commit_field(fid, arr) == want
        tt = mz.MerkleTree(fid, arr)
        assert tt.root() == want
        tt.close()


def test_ragged_one_shot_device_commit(mz):
    import torch
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 777
    arr = _edge_vector(M128, 42, n)
    d = torch.from_numpy(arr.view(np.int64).reshape(-1).copy()).to(dev)
    root, ln = (ctypes.c_uint8 * 48)(), ctypes.c_size_t()
    assert L.mzk_merkle_commit_field_dev(M128, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), root, ctypes.c_size_t(48), ctypes.byref(ln), st) == 0
    assert bytes(root[:ln.value]) == orc.merkle_commit_ref(orc.field_leaves(M128, arr))


def test_device_resident_fri_commit_loop(mz):
    """FRI::commit (fri.rs:144-209) with the codeword never leaving HBM: per round Merkle root -> (host transcript)
    -> alpha -> split-and-fold; the transcript is stood in for by alpha = SHA3(root) mod p on both sides."""
    import torch
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lg = 14
    n = 1 << lg
    cw = orc.synth_vector(M128, 55, n)
    omega, offset = orc.m128_root(lg), orc.M128_GEN
    p = orc.MOD[M128]
    d_cw = torch.from_numpy(cw.view(np.int64).reshape(-1).copy()).to(dev)
    cpu = cw
    for r in range(6):
        root = (ctypes.c_uint8 * 48)()
        ln = ctypes.c_size_t()
        rc = L.mzk_merkle_commit_field_dev(M128, ctypes.c_void_p(d_cw.data_ptr()), ctypes.c_size_t(cpu.shape[0]), root, ctypes.c_size_t(48),
                                           ctypes.byref(ln), st)
        assert rc == 0, L.mzk_last_error().decode()
        want_root = orc.merkle_commit_ref(orc.field_leaves(M128, cpu))
        assert bytes(root[:ln.value]) == want_root
        alpha = int.from_bytes(hashlib.sha3_256(want_root).digest(), "little") % p
        a, o, w = (orc.to_limbs([v], 2) for v in (alpha, offset, omega))
        d_next = torch.zeros(cpu.shape[0], dtype=torch.int64, device=dev)        # (n/2) * 2 limbs
        rc = L.mzk_fri_fold_dev(M128, ctypes.c_void_p(d_cw.data_ptr()), ctypes.c_size_t(cpu.shape[0]), a.ctypes.data_as(ctypes.c_void_p),
                                o.ctypes.data_as(ctypes.c_void_p), w.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_next.data_ptr()), st)
        assert rc == 0, L.mzk_last_error().decode()
        torch.cuda.synchronize()
        cpu = orc.fri_fold_ref(M128, cpu, alpha, offset, omega)
        assert np.array_equal(d_next.cpu().numpy().view(np.uint64).reshape(-1, 2), cpu)
        d_cw = d_next
        omega, offset = omega * omega % p, offset * offset % p


def test_large_codeword_root(mz):
    """2^20 M128 elements (BASELINE configs[2] size): root vs the oracle's literal recursion (~1 s of CPU)."""
    arr = orc.synth_vector(M128, 1234, 1 << 20)
    assert mz.merkle_commit_field(M128, arr) == orc.merkle_commit_ref(orc.field_leaves_fast(M128, arr))


@pytest.mark.parametrize("fid,lg,rounds", [(M128, 12, 7), (FR, 9, 4), (M128, 3, 4), (M128, 0, 1)])
def test_fri_commit_entry_point_matches_oracle_replay(mz, fid, lg, rounds):
    """mzk_fri_commit (fri.rs:144-209): roots, every round's codeword, the challenge callback protocol (one call per
    round, alpha ignored on the last) and the omega/offset squaring; (M128, 3, 4) ends on a one-element codeword whose
    "root" is the leaf itself (merkle.rs:17-19)."""
    n = 1 << lg
    p = orc.MOD[fid]
    nl = orc.LIMBS[fid]
    cw = orc.synth_vector(fid, 900 + lg, n)
    omega = orc.root_of(fid, lg) if lg else 1
    offset = orc.M128_GEN if fid == M128 else 5
    calls = []

    def challenge(rnd, last, root):
        calls.append((rnd, last, root))
        return None if last else int.from_bytes(hashlib.sha3_256(root + bytes([rnd])).digest(), "little") % p

    cws, roots = mz.fri_commit(fid, cw, omega, offset, rounds, challenge)
    assert [c[0] for c in calls] == list(range(rounds)) and [c[1] for c in calls] == [False] * (rounds - 1) + [True]
    cur, om, of = cw, omega, offset
    for r in range(rounds):
        want_root = orc.merkle_commit_ref(orc.field_leaves(fid, cur))
        assert roots[r] == want_root == calls[r][2], r
        assert np.array_equal(cws[r], cur), r
        if r == rounds - 1:
            break
        alpha = int.from_bytes(hashlib.sha3_256(want_root + bytes([r])).digest(), "little") % p
        cur = orc.fri_fold_ref(fid, cur, alpha, of, om)
        om, of = om * om % p, of * of % p


def test_fri_commit_contract_violations(mz):
    cw = orc.synth_vector(M128, 1, 8)
    with pytest.raises(mz.MzkError) as e:
        mz.fri_commit(M128, cw, orc.root_of(M128, 3), 3, 5, lambda *a: 1)      # 5 rounds on 8 elements
    assert e.value.code == -5
    with pytest.raises(mz.MzkError) as e:
        mz.fri_commit(M128, cw[:6], orc.root_of(M128, 3), 3, 2, lambda *a: 1)
    assert e.value.code == -2
    with pytest.raises(mz.MzkError) as e:
        mz.fri_commit(M128, cw, orc.root_of(M128, 3), 3, 2, lambda *a: orc.MOD[M128])   # non-canonical challenge
    assert e.value.code == -6


@pytest.mark.parametrize("fid,n", [(M128, 1), (M128, 2), (M128, 64), (FR, 256), (M128, 1000)])
def test_signed_leaves_reproduce_bincode_of_unsanitized_elements(mz, fid, n):
    """SURVEY F6 / VERDICT r01 weak #8: the reference's round-0 FRI root hashes bincode(c) of UNSANITIZED elements
    (fri.rs:160-166); a negative BigInt serialises as Sign::Minus + magnitude.  The *_signed entry points take
    (magnitude, negative) and must hash exactly those bytes; paths too."""
    nl = orc.LIMBS[fid]
    mag = _edge_vector(fid, 600 + n, n)
    rnd = random.Random(n)
    neg = np.array([rnd.randrange(2) for _ in range(n)], dtype=np.uint8)
    leaves = [orc.bincode_field_signed(v, nl, s) for v, s in zip(orc.from_limbs(mag), neg)]
    assert any(l[0] == 0xff for l in leaves) or n < 4
    t = mz.MerkleTree(fid, mag, negative=neg)
    assert t.root() == orc.merkle_commit_ref(leaves)
    if n >= 2:
        for idx in {0, 1, n - 1, n // 2}:
            want = orc.merkle_open_ref(idx, leaves)
            if want is not None:
                assert t.open(idx) == want
    t.close()


def test_fri_commit_signed_round0_root_then_canonical_folds(mz):
    fid, lg, rounds = M128, 8, 4
    n, p = 1 << lg, orc.MOD[M128]
    mag = _edge_vector(fid, 777, n)
    neg = np.array([(i * 7) % 3 == 0 for i in range(n)], dtype=np.uint8)
    omega, offset = orc.root_of(fid, lg), orc.M128_GEN

    def challenge(rnd, last, root):
        return None if last else int.from_bytes(hashlib.sha3_256(root + bytes([rnd])).digest(), "little") % p

    cws, roots = mz.fri_commit(fid, mag, omega, offset, rounds, challenge, negative=neg)
    leaves0 = [orc.bincode_field_signed(v, 2, s) for v, s in zip(orc.from_limbs(mag), neg)]
    assert roots[0] == orc.merkle_commit_ref(leaves0)
    canon = orc.to_limbs([(p - v) % p if s else v for v, s in zip(orc.from_limbs(mag), neg)], 2)
    assert np.array_equal(cws[0], canon)
    cur, om, of = canon, omega, offset
    for r in range(rounds - 1):
        alpha = int.from_bytes(hashlib.sha3_256(roots[r] + bytes([r])).digest(), "little") % p
        cur = orc.fri_fold_ref(fid, cur, alpha, of, om)
        om, of = om * om % p, of * of % p
        assert np.array_equal(cws[r + 1], cur)
        assert roots[r + 1] == orc.merkle_commit_ref(orc.field_leaves(fid, cur))


def test_fri_commit_callback_exception_aborts_instead_of_folding_with_zero(mz):
    """an exception in the transcript callback must surface, not fold with alpha = 0."""
    cw = orc.synth_vector(M128, 3, 16)

    def challenge(rnd, last, root):
        if rnd == 1:
            raise KeyError("transcript exploded")
        return 5

    with pytest.raises(KeyError):
        mz.fri_commit(M128, cw, orc.root_of(M128, 4), orc.M128_GEN, 3, challenge)
    with pytest.raises(ValueError):
        mz.fri_commit(M128, cw, orc.root_of(M128, 4), orc.M128_GEN, 3, lambda *a: None)      # forgot to return alpha


@pytest.mark.parametrize("fid", [0, 1])
def test_batched_commit_equals_single_commits(fid):
    """mzk_merkle_commit_field_batch: the provers' per-register loop (fast_stark.rs:231-243) as one call -- every root equal
    to Merkle::commit of that codeword alone: two-leaf trees, trees that end inside the one-workgroup tail, more trees
    than the tail holds nodes, a batch count that is not a power of two; other leaf counts are refused."""
    import myzkp_amd as mz
    mz.init(0)
    nl = mz.LIMBS[fid]
    for n, batch in ((2, 5), (4, 1), (8, 3), (1024, 16), (1 << 14, 7), (4, 600), (1 << 12, 33)):
        cws = np.stack([orc.synth_vector(fid, 700 + 11 * k + n, n) for k in range(batch)])
        roots = mz.merkle_commit_field_batch(fid, cws)
        assert len(roots) == batch
        for k in sorted({0, 1 % batch, batch // 2, batch - 1}):
            assert roots[k] == mz.merkle_commit_field(fid, cws[k]), (n, batch, k)
    assert mz.merkle_commit_field_batch(fid, np.zeros((0, 8, nl), dtype=np.uint64)) == []
    for bad in (1, 6):
        with pytest.raises(mz.MzkError):
            mz.merkle_commit_field_batch(fid, np.zeros((2, bad, nl), dtype=np.uint64))


@pytest.mark.parametrize("fid,lg", [(M128, 1), (M128, 2), (M128, 12), (FR, 5), (FR, 11)])
def test_open_many_equals_single_openings_and_the_oracle(mz, fid, lg):
    """mzk_merkle_open_batch: the FRI query phase's openings (fri.rs:211-260) in one gather -- every path equal to
    Merkle::open (merkle.rs:27-46) by the oracle and to the single call; repeated indices, the signed-leaf tree, zero
    openings; byte-leaf and ragged trees are refused."""
    n = 1 << lg
    arr = _edge_vector(fid, 900 + lg, n)
    leaves = orc.field_leaves(fid, arr)
    t = mz.MerkleTree(fid, arr)
    rnd = random.Random(100 + lg)
    idx = [0, 1, n - 1, n // 2, 0] + [rnd.randrange(n) for _ in range(40)]
    paths = t.open_many(idx)
    assert len(paths) == len(idx)
    for q, i in enumerate(idx):
        assert paths[q] == orc.merkle_open_ref(i, leaves), (q, i)
    assert paths[3] == t.open(n // 2)
    assert t.open_many([]) == []
    with pytest.raises(mz.MzkError):
        t.open_many([n])
    t.close()
    if lg >= 2:
        neg = np.array([(k % 3) == 0 for k in range(n)], dtype=np.uint8)
        ts = mz.MerkleTree(fid, arr, negative=neg)
        ps = ts.open_many([1, n - 2])
        assert ps == [ts.open(1), ts.open(n - 2)]
        ts.close()
        tb = mz.MerkleTree(leaves=[b"a", b"bb", b"ccc", b"dddd"])
        with pytest.raises(mz.MzkError):
            tb.open_many([1])
        tb.close()


@pytest.mark.parametrize("fid,lg,rounds,signed", [(M128, 10, 6, False), (FR, 6, 4, False), (M128, 3, 4, False), (M128, 8, 5, True)])
def test_fri_commit_keeps_the_round_trees_for_the_query_phase(mz, fid, lg, rounds, signed):
    """mzk_fri_commit_keep_trees: same codewords and roots as the plain loop, plus one device-resident tree per round whose
    openings are Merkle::open on that round's codeword (fri.rs:211-260 opens a, b from round r and c from round r + 1);
    a one-element round has no tree; round 0 of the signed form opens the unsanitized leaves."""
    n = 1 << lg
    p = orc.MOD[fid]
    cw = orc.synth_vector(fid, 1300 + lg, n)
    neg = np.array([(k % 5) == 1 for k in range(n)], dtype=np.uint8) if signed else None
    omega, offset = orc.root_of(fid, lg), (orc.M128_GEN if fid == M128 else 5)

    def challenge(rnd, last, root):
        return None if last else int.from_bytes(hashlib.sha3_256(root + bytes([rnd])).digest(), "little") % p

    cws0, roots0 = mz.fri_commit(fid, cw, omega, offset, rounds, challenge, negative=neg)
    cws, roots, trees = mz.fri_commit(fid, cw, omega, offset, rounds, challenge, negative=neg, keep_trees=True)
    assert roots == roots0 and all(np.array_equal(a, b) for a, b in zip(cws, cws0))
    rnd = random.Random(lg)
    for r in range(rounds):
        m = n >> r
        if m < 2:
            assert trees[r] is None
            continue
        if r == 0 and signed:
            leaves = [orc.bincode_field_signed(v, orc.LIMBS[fid], s) for v, s in zip(orc.from_limbs(cw), neg)]
        else:
            leaves = orc.field_leaves(fid, cws[r])
        assert trees[r].root() == roots[r]
        idx = [rnd.randrange(m) for _ in range(9)] + [0, m - 1]
        paths = trees[r].open_many(idx)
        for q, i in enumerate(idx):
            assert paths[q] == orc.merkle_open_ref(i, leaves), (r, i)
            assert orc.merkle_verify_ref(roots[r], i, paths[q], leaves[i])
    for t in trees:
        if t is not None:
            t.close()


def test_merkle_open_multi_equals_tree_by_tree(mz):
    """mzk_merkle_open_multi: the openings of several trees (different sizes and fields' worth of leaves, repeated and empty index
    lists, a skipped NULL tree, signed leaves) in one call == MerkleTree.open_many tree by tree == the oracle's Merkle::open."""
    import random
    rnd = random.Random(5)
    trees, lists = [], []
    for fid, lg in ((orc.M128, 10), (orc.M128, 4), (orc.FR, 7), (orc.M128, 1)):
        x = orc.synth_vector(fid, 50 + lg, 1 << lg)
        trees.append((fid, x, mz.MerkleTree(fid, x)))
        lists.append([rnd.randrange(1 << lg) for _ in range(rnd.choice([1, 5, 40]))])
    lists[1] = []                                   # nothing to open in this tree
    lists[0] += [lists[0][0], 0, (1 << 10) - 1]     # repeated index, first and last leaf
    handles = [t for _, _, t in trees] + [None]
    got = mz.merkle_open_multi(handles, lists + [[]])
    assert len(got) == 5 and got[1] == [] and got[4] == []
    for (fid, x, t), idx, paths in zip(trees, lists, got):
        assert paths == t.open_many(idx) if idx else paths == []
        leaves = orc.field_leaves(fid, x)
        for i, p in zip(idx, paths):
            assert [bytes(e) for e in p] == [bytes(e) for e in orc.merkle_open_ref(i, leaves)]
            assert orc.merkle_verify_ref(t.root(), i, p, leaves[i])
    # out-of-range index in the second non-empty tree: refused, naming the tree
    with pytest.raises(mz.MzkError) as e:
        mz.merkle_open_multi([trees[0][2], trees[2][2]], [[1], [1 << 7]])
    assert e.value.code == -5
    # byte-leaf trees are opened one by one (as with open_batch)
    bt = mz.MerkleTree(leaves=[b"leaf%d" % i for i in range(4)])
    with pytest.raises(mz.MzkError) as e:
        mz.merkle_open_multi([bt], [[1]])
    assert e.value.code == -1
    bt.close()
    for _, _, t in trees:
        t.close()


def test_fri_commit_device_resident_and_values_from_the_trees(mz):
    """mzk_fri_commit_keep_trees_dev (initial codeword in HBM) with codewords_out = NULL: roots identical to the host form, and the
    query phase takes everything from the trees -- mzk_merkle_leaves returns exactly the codeword elements (all of the last
    codeword, sampled positions elsewhere; signs of an unsanitized round 0), mzk_merkle_open_multi the paths."""
    import torch
    fid, lg, rounds = orc.M128, 12, 6
    n = 1 << lg
    p = orc.MOD[fid]
    cw = orc.synth_vector(fid, 321, n)
    omega, offset = orc.root_of(fid, lg), orc.M128_GEN
    seen = []

    def challenge(rnd, last, root):
        seen.append(root)
        return None if last else (int.from_bytes(root[:15], "little") + rnd) % p

    cws, roots, trees = mz.fri_commit(fid, cw, omega, offset, rounds, challenge, keep_trees=True)
    for t in trees:
        t.close()
    d = torch.from_numpy(cw.view(np.int64).reshape(-1).copy()).to("cuda:0")
    torch.cuda.synchronize()
    none, roots_d, trees_d = mz.fri_commit(fid, None, omega, offset, rounds, challenge, keep_trees=True, codewords=False, device_ptr=d.data_ptr(), n=n)
    assert none is None and roots_d == roots
    for r, t in enumerate(trees_d):
        m = n >> r
        idx = list(range(m)) if r == rounds - 1 else [0, 1, m // 2, m - 1, 17 % m, 17 % m]
        assert np.array_equal(t.leaves(idx), cws[r][idx])
    opened = mz.merkle_open_multi(trees_d, [[3, 5]] * rounds)
    for r, t in enumerate(trees_d):
        leaves = orc.field_leaves(fid, cws[r])
        assert [bytes(e) for e in opened[r][0]] == [bytes(e) for e in orc.merkle_open_ref(3, leaves)]
        assert orc.merkle_verify_ref(roots[r], 5, opened[r][1], leaves[5])
    # out-of-range position
    with pytest.raises(mz.MzkError) as e:
        trees_d[0].leaves([n])
    assert e.value.code == -5
    for t in trees_d:
        t.close()
    # signed round 0: magnitudes and flags come back as given
    neg = (np.arange(64) % 3 == 0).astype(np.uint8)
    mag = orc.synth_vector(fid, 9, 64)
    _, _, ts = mz.fri_commit(fid, mag, orc.root_of(fid, 6), offset, 2, challenge, negative=neg, keep_trees=True, codewords=False)
    got, sg = ts[0].leaves(list(range(64)), with_sign=True)
    assert np.array_equal(got, mag) and np.array_equal(sg, neg)
    for t in ts:
        t.close()
