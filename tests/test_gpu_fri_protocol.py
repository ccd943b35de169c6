"""FRI prove / verify end to end, the reference's own test_fri_field (zkstark/fri.rs:495-545) re-expressed on the GPU entry
points: degree 63, expansion factor 4, 17 colinearity tests, offset = the M128 generator, omega from get_nth_root_of_m128.

  prover    codeword = polynomial.eval_domain(omega^i)            -> mzk_ntt                       (fri.rs:521-522)
            FRI::commit: Merkle roots + split-and-fold rounds      -> mzk_fri_commit_keep_trees     (fri.rs:144-209)
            FRI::reveal: a / b / c values and authentication paths -> mzk_merkle_open_multi         (fri.rs:211-260; all rounds, one call)
  verifier  fri.rs:262-400 restated with Python integers and the oracle's Merkle::verify: last-codeword root, its degree,
            colinearity of every (a, b, c) triple, all authentication paths, and the revealed points equal polynomial.eval
            (fri.rs:527-529); a codeword corrupted as in fri.rs:531-538 must be rejected.

The proof stream (FiatShamirTransformer, Blake2b index sampling, F::sample) is host control plane in the reference and out of
this path's scope: a SHA3 transcript stands in for it on BOTH sides, so prover and verifier derive the same challenges."""
import hashlib
import numpy as np
import pytest
import orc
from orc import M128

pytestmark = pytest.mark.gpu
P = orc.MOD[M128]
GEN = 85408008396924667383611388730472331217          # fri.rs:506-508 (order 2^119)


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd as m
    m.init(0)
    return m


def num_rounds(domain_length, expansion_factor, num_colinearity_tests):      # fri.rs:86-97
    n, r = domain_length, 0
    while n > expansion_factor and 4 * num_colinearity_tests < n:
        n //= 2
        r += 1
    return r


def h(*parts):
    return hashlib.sha3_256(b"".join(parts)).digest()


def sample_indices(seed, size, reduced_size, number):                        # fri.rs:40-62, SHA3 for Blake2b
    indices, reduced, counter = [], [], 0
    while len(indices) < number:
        idx = int.from_bytes(h(seed, counter.to_bytes(8, "little")), "little") % size
        counter += 1
        if idx % reduced_size not in reduced:
            indices.append(idx)
            reduced.append(idx % reduced_size)
    return indices


def prove(mz, codeword, omega, offset, expansion, tests):
    n = codeword.shape[0]
    rounds = num_rounds(n, expansion, tests)
    roots_seen = []

    def challenge(rnd, last, root):                                          # the transcript: push the root, then sample alpha
        roots_seen.append(root)
        return None if last else int.from_bytes(h(*roots_seen), "little") % P

    codewords, roots, trees = mz.fri_commit(M128, codeword, omega, offset, rounds, challenge, keep_trees=True)
    assert roots == roots_seen
    last_leaves = b"".join(orc.field_leaves(M128, codewords[-1]))
    top = sample_indices(h(*roots, last_leaves), codewords[1].shape[0], codewords[-1].shape[0], tests)   # fri.rs:117-123
    # query phase (fri.rs:127-137 + reveal): the a / b indices of every round in that round's tree, the c indices in the next
    # round's -- all openings of all rounds in ONE call (mzk_merkle_open_multi); tree i serves b_(i) + a_(i) and c_(i-1)
    per_round, indices = [], list(top)
    for i in range(len(codewords) - 1):
        half = codewords[i].shape[0] // 2
        indices = [idx % half for idx in indices]
        per_round.append((list(indices), [idx + half for idx in indices]))
    wanted = [[] for _ in codewords]
    for i, (a, b) in enumerate(per_round):
        wanted[i] += a + b
        wanted[i + 1] += a                                                    # c indices = a indices, in the next codeword
    opened = mz.merkle_open_multi(trees, wanted)
    if len(codewords) > 1:                                                    # the same paths as tree by tree
        assert opened[0] == trees[0].open_many(wanted[0]) and opened[1] == trees[1].open_many(wanted[1])
    layers = []
    for i, (a, b) in enumerate(per_round):
        cur, nxt = orc.from_limbs(codewords[i]), orc.from_limbs(codewords[i + 1])
        skip = tests if i > 0 else 0                                         # tree i's list starts with round i-1's c paths
        pa, pb = opened[i][skip:skip + tests], opened[i][skip + tests:skip + 2 * tests]
        pc = opened[i + 1][:tests] if i + 1 > 0 else []
        layers.append({"a": ([cur[j] for j in a], pa), "b": ([cur[j] for j in b], pb), "c": ([nxt[j] for j in a], pc)})
    for t in trees:
        if t is not None:
            t.close()
    return {"top_level_indices": top, "last_codeword": codewords[-1], "merkle_roots": roots, "revealed_layers": layers}


def leaf(v):
    return orc.bincode_field(v, 2)


def verify(proof, omega, offset, domain_length, expansion, tests, points):    # fri.rs:262-400
    rounds = num_rounds(domain_length, expansion, tests)
    roots = proof["merkle_roots"]
    alphas = [int.from_bytes(h(*roots[:r + 1]), "little") % P for r in range(len(roots))]
    last = proof["last_codeword"]
    if orc.merkle_commit_field_ref(M128, last) != roots[-1]:
        return False
    degree = last.shape[0] // expansion - 1
    last_omega, last_offset = pow(omega, 1 << (rounds - 1), P), pow(offset, 1 << (rounds - 1), P)
    assert pow(last_omega, -1, P) == pow(last_omega, last.shape[0] - 1, P), "omega does not have right order"
    rc, scaled = orc.ntt_fast(M128, last_omega, np.ascontiguousarray(last), True)      # interpolant on last_offset * last_omega^i
    assert rc == 0
    inv = pow(last_offset, -1, P)
    coef = [c * pow(inv, i, P) % P for i, c in enumerate(orc.from_limbs(scaled))]
    if any(coef[degree + 1:]):
        return False
    last_leaves = b"".join(orc.field_leaves(M128, last))
    top = sample_indices(h(*roots, last_leaves), domain_length >> 1, domain_length >> (rounds - 1), tests)
    if top != proof["top_level_indices"]:
        return False
    for r in range(rounds - 1):
        c_idx = [i % (domain_length >> (r + 1)) for i in top]
        a_idx, b_idx = c_idx, [i + (domain_length >> (r + 1)) for i in c_idx]
        L = proof["revealed_layers"][r]
        for s in range(tests):
            ay, by, cy = L["a"][0][s], L["b"][0][s], L["c"][0][s]
            if r == 0:
                points += [(a_idx[s], ay), (b_idx[s], by)]
            ax, bx, cx = offset * pow(omega, a_idx[s], P) % P, offset * pow(omega, b_idx[s], P) % P, alphas[r]
            if (by - ay) * (cx - ax) % P != (cy - ay) * (bx - ax) % P:                 # the interpolant through the three has degree <= 1
                return False
        for s in range(tests):
            if not orc.merkle_verify_ref(roots[r], a_idx[s], L["a"][1][s], leaf(L["a"][0][s])):
                return False
            if not orc.merkle_verify_ref(roots[r], b_idx[s], L["b"][1][s], leaf(L["b"][0][s])):
                return False
            if not orc.merkle_verify_ref(roots[r + 1], c_idx[s], L["c"][1][s], leaf(L["c"][0][s])):
                return False
        omega, offset = omega * omega % P, offset * offset % P
    return True


@pytest.mark.parametrize("degree,expansion,tests", [(63, 4, 17), (4095, 4, 17), (1023, 8, 10)])
def test_fri_field(mz, degree, expansion, tests):
    n = (degree + 1) * expansion
    lg = n.bit_length() - 1
    assert 1 << lg == n
    omega = orc.m128_root(lg)
    assert mz.root_of_unity(M128, lg) == omega and pow(omega, n, P) == 1 and pow(omega, n // 2, P) != 1
    coef = list(range(degree + 1))                                                       # fri.rs:514-518
    codeword = mz.ntt(M128, omega, orc.to_limbs(coef + [0] * (n - degree - 1), 2))      # eval_domain(omega^i)
    proof = prove(mz, codeword, omega, GEN, expansion, tests)
    assert len(proof["merkle_roots"]) == num_rounds(n, expansion, tests)
    points = []
    assert verify(proof, omega, GEN, n, expansion, tests, points)
    assert len(points) == 2 * tests
    for x, y in points:                                                                  # fri.rs:527-529
        assert sum(c * pow(omega, x * i, P) for i, c in enumerate(coef)) % P == y
    # fri.rs:531-538: a codeword that is far from low degree must be rejected
    bad = codeword.copy()
    bad[:degree // 3] = orc.to_limbs([1], 2)[0]
    assert not verify(prove(mz, bad, omega, GEN, expansion, tests), omega, GEN, n, expansion, tests, [])
    # and a proof with one revealed value changed fails its authentication path or the colinearity check
    L0 = proof["revealed_layers"][0]
    L0["a"][0][3] = (L0["a"][0][3] + 1) % P
    assert not verify(proof, omega, GEN, n, expansion, tests, [])
