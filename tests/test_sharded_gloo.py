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
