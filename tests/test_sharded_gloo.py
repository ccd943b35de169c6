"""The N>1 path on CPU: world_size-2 gloo.  The sharding / all-gather / fold logic of
myzkp_amd/sharded.py is exercised with the oracle standing in for the per-rank GPU compute (tests may
use the oracle); the real run swaps in mzk_msm_g1_bn254_partial_dev / mzk_g1_fold_partials_dev."""
import os, socket, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import orc
    from myzkp_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = orc.synth_vector(orc.FR, 42, n, 1)
    p = orc.synth_points(43, n, 1)
    lo, hi = sharded.shard_range(n, rank, world)

    def local_partial():
        pt = orc.msm_fast(s[lo:hi], p[lo:hi], 1)          # stand-in for the GPU partial
        return torch.from_numpy(orc.pts_to_arr([pt]).view(np.int64).reshape(-1).copy())

    def fold(records):
        acc = (0, 0)
        for r in range(records.shape[0]):
            acc = orc.ec_add(0, acc, orc.arr_to_pts(records[r].numpy().view(np.uint64))[0])
        return acc

    got = sharded.sharded_msm(local_partial, fold)
    want = orc.msm_fast(s, p, 1)
    q.put((rank, got == want, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 101])
def test_sharded_msm_world2_gloo(n):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    spans = sorted((lo, hi) for _, _, lo, hi in res)
    assert spans[0][0] == 0 and spans[-1][1] == n and spans[0][1] == spans[1][0]   # contiguous cover


def test_shard_range_balanced():
    sys.path.insert(0, ROOT)
    from myzkp_amd import sharded
    for n in (0, 1, 7, 8, 1 << 20, (1 << 24) + 3):
        for w in (1, 2, 4, 8):
            spans = [sharded.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- open_kzg's quotient sharded over the ranks: one 32-byte value per rank exchanged ------------------------------
class _OracleOpenOps:
    """The two local passes of sharded.sharded_open_quotient with Python integers standing in for the GPU (mzk_kzg_open_slice_value_dev /
    mzk_kzg_open_slice_quotient_dev), the gather over gloo."""

    def __init__(self, orc, sharded, p):
        self.orc, self.sharded, self.p = orc, sharded, p

    def slice_value(self, coef, u):
        acc = 0
        for c in reversed(coef):
            acc = (acc * u + c) % self.p
        return acc

    def gather_values(self, value):
        t = torch.from_numpy(self.orc.to_limbs([value], 4).view(np.int64).reshape(-1).copy())
        recs = self.sharded.all_gather_values(t)
        return self.orc.from_limbs(recs.numpy().view(np.uint64).reshape(-1, 4))

    def slice_quotient(self, coef, u, carry):
        b, out = carry, [0] * len(coef)
        for i in range(len(coef) - 1, -1, -1):
            out[i] = b                       # q[lo + i] = b[lo + i + 1]
            b = (coef[i] + u * b) % self.p
        return out


def _open_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import orc
    from myzkp_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = orc.P_FR
    coef = orc.from_limbs(orc.synth_vector(orc.FR, 51, n, 1))
    srs = orc.synth_points(52, n, 1)
    u = orc.from_limbs(orc.synth_vector(orc.FR, 53, 1, 1))[0]
    lo, hi = sharded.shard_range(n, rank, world)
    ops = _OracleOpenOps(orc, sharded, p)
    y, qs = sharded.sharded_open_quotient(ops, coef[lo:hi], n, u, p, rank, world)
    # the rank commits its slice of q against ITS powers; the partials are gathered and folded like every sharded MSM
    pt = orc.msm_fast(orc.to_limbs(qs, 4), srs[lo:hi], 1) if hi > lo else (0, 0)
    recs = sharded.all_gather_partials(torch.from_numpy(orc.pts_to_arr([pt]).view(np.int64).reshape(-1).copy()))
    w = (0, 0)
    for r in range(recs.shape[0]):
        w = orc.ec_add(0, w, orc.arr_to_pts(recs[r].numpy().view(np.uint64))[0])
    want_y, want_w = orc.kzg_open_ref(orc.to_limbs(coef, 4), u, srs)
    q.put((rank, y == want_y, w == want_w, len(qs) == hi - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(97, 2), (8, 2), (3, 2), (65, 3)])
def test_sharded_open_quotient_gloo(n, world):
    """kzg.rs:61-72 over the ranks: y and the folded witness equal the oracle's literal open_kzg (orc.kzg_open_ref: evaluation, synthetic
    division with the reference's trimming, the MSM by affine double-and-add) -- ragged slices, a slice of one coefficient, three ranks."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_open_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b and c for _, a, b, c in res), res


def test_open_carries_is_the_suffix_recurrence():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import orc
    from myzkp_amd import sharded
    p = orc.P_FR
    coef = orc.from_limbs(orc.synth_vector(orc.FR, 77, 41, 1))
    u = 123456789123456789
    b = [0] * 42
    for i in range(40, -1, -1):
        b[i] = (coef[i] + u * b[i + 1]) % p
    for world in (1, 2, 5, 8):
        spans = [sharded.shard_range(41, g, world) for g in range(world)]
        values = [sum(coef[lo + t] * pow(u, t, p) for t in range(hi - lo)) % p for lo, hi in spans]
        y, carries = sharded.open_carries(values, [hi - lo for lo, hi in spans], u, p)
        assert y == b[0] and carries == [b[hi] for _, hi in spans]
