"""CPU: the Merkle / SHA3 / leaf-byte oracle (oracle/mzk_oracle_merkle.c) against independent anchors.

SHA3-256 is pinned by the FIPS 202 known answers and by hashlib on random inputs; Merkle::commit/open/verify are
pinned against a pure-Python restatement of merkle.rs:15-67 built on hashlib and by the reference's own test
(merkle.rs:76-93: leaf1..leaf4, index 2).  The bincode layout is restated from the published num-bigint 0.4 /
bincode 1.3 formats (PARITY UNPINNED: neither crate is under /root/reference and no golden root exists)."""
import hashlib, os, random, struct
import numpy as np
import pytest
import orc
from orc import FR, M128


def test_sha3_256_fips202_known_answers():
    assert orc.sha3_256(b"").hex() == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a"
    assert orc.sha3_256(b"abc").hex() == "3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532"
    assert orc.sha3_256(b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq").hex() == \
        "41c0dba2a9d6240849100376a8235e2c82e1b9998a999e21db32dd97496d3376"


def test_sha3_256_matches_hashlib_around_the_rate_boundary():
    rnd = random.Random(1)
    for n in list(range(0, 20)) + [63, 64, 65, 82, 134, 135, 136, 137, 271, 272, 273, 1000]:
        d = bytes(rnd.randrange(256) for _ in range(n))
        assert orc.sha3_256(d) == hashlib.sha3_256(d).digest(), n


def py_bincode(v):
    """bincode 1.3 of (Sign as i8, Vec<u32> digits): see the oracle header."""
    digits = []
    x = v
    while x:
        digits.append(x & 0xffffffff)
        x >>= 32
    return struct.pack("<bQ", 1 if v else 0, len(digits)) + b"".join(struct.pack("<I", d) for d in digits)


def test_bincode_layout_examples():
    assert orc.bincode_field(0, 2) == bytes([0]) + bytes(8)
    assert orc.bincode_field(5, 2) == bytes([1, 1, 0, 0, 0, 0, 0, 0, 0, 5, 0, 0, 0])
    assert orc.bincode_field(1 << 32, 4) == bytes([1, 2, 0, 0, 0, 0, 0, 0, 0]) + bytes(4) + bytes([1, 0, 0, 0])
    # Sign::Minus = -1i8 = 0xff, magnitude digits follow (an element the reference left unsanitized, field.rs:98-110)
    assert orc.bincode_field_signed(5, 2, True) == bytes([0xff, 1, 0, 0, 0, 0, 0, 0, 0, 5, 0, 0, 0])
    assert orc.bincode_field_signed((1 << 64) + 7, 4, True) == bytes([0xff, 3, 0, 0, 0, 0, 0, 0, 0, 7, 0, 0, 0]) + bytes(4) + bytes([1, 0, 0, 0])
    assert orc.bincode_field_signed(0, 2, True) == bytes(9)          # BigInt has no negative zero
    assert orc.bincode_field_signed(9, 2, False) == orc.bincode_field(9, 2)
    rnd = random.Random(2)
    for fid, nl in ((FR, 4), (M128, 2)):
        for bits in (1, 31, 32, 33, 64, 65, 96, 127, 128, 200, 253):
            v = rnd.getrandbits(bits) % orc.MOD[fid]
            assert orc.bincode_field(v, nl) == py_bincode(v)
        assert orc.bincode_field(orc.MOD[fid] - 1, nl) == py_bincode(orc.MOD[fid] - 1)


def py_commit(leaves):          # merkle.rs:15-25
    if len(leaves) == 1:
        return leaves[0]
    mid = len(leaves) // 2
    return hashlib.sha3_256(py_commit(leaves[:mid]) + py_commit(leaves[mid:])).digest()


def py_open(index, leaves):     # merkle.rs:28-46
    if len(leaves) == 2:
        return [leaves[1 - index]]
    mid = len(leaves) // 2
    if index < mid:
        return py_open(index, leaves[:mid]) + [py_commit(leaves[mid:])]
    return py_open(index - mid, leaves[mid:]) + [py_commit(leaves[:mid])]


def test_reference_merkle_test_leaf1_to_leaf4():
    """merkle.rs:76-93"""
    leaves = [b"leaf1", b"leaf2", b"leaf3", b"leaf4"]
    root = orc.merkle_commit_ref(leaves)
    assert root == py_commit(leaves)
    proof = orc.merkle_open_ref(2, leaves)
    assert proof == py_open(2, leaves)
    assert orc.merkle_verify_ref(root, 2, proof, leaves[2])
    assert not orc.merkle_verify_ref(root, 2, proof, leaves[3])


@pytest.mark.parametrize("n", [1, 2, 4, 8, 64, 3, 5, 6])
def test_commit_open_verify_match_python_restatement(n):
    rnd = random.Random(n)
    leaves = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 9, 25, 41, 70, 140, 300]))) for _ in range(n)]
    root = orc.merkle_commit_ref(leaves)
    assert root == py_commit(leaves)
    if n >= 2 and n & (n - 1) == 0:
        for idx in range(n):
            path = orc.merkle_open_ref(idx, leaves)
            assert path == py_open(idx, leaves)
            assert orc.merkle_verify_ref(root, idx, path, leaves[idx])


def test_field_codeword_leaves_and_root():
    arr = orc.synth_vector(M128, 77, 16)
    arr[3] = 0
    arr[4] = orc.to_limbs([7], 2)[0]
    leaves = orc.field_leaves(M128, arr)
    assert leaves == [py_bincode(v) for v in orc.from_limbs(arr)]
    assert len(leaves[3]) == 9 and len(leaves[4]) == 13
    assert orc.merkle_commit_ref(leaves) == py_commit(leaves)


@pytest.mark.parametrize("n", [3, 5, 6, 7, 9, 12, 13, 21, 100])
def test_ragged_open_terminates_exactly_where_the_reference_does(n):
    """merkle.rs:28-46 on a ragged slice: the Python restatement hits RecursionError exactly where the oracle reports
    non-termination (a one-leaf slice has mid = 0); everywhere else the paths agree."""
    rnd = random.Random(n)
    leaves = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 5, 32, 33, 140]))) for _ in range(n)]
    assert orc.merkle_commit_ref(leaves) == py_commit(leaves)
    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(200)
    try:
        for idx in range(n):
            got = orc.merkle_open_ref(idx, leaves)
            try:
                want = py_open(idx, leaves)
            except RecursionError:
                want = None
            assert got == want, idx
    finally:
        sys.setrecursionlimit(old)
