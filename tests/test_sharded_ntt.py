"""One transform sharded over W ranks (myzkp_amd/sharded.py, SURVEY 8e four-step layout) on CPU: the schedule -- slices,
transposes, twiddles, the three layouts, forward and inverse -- with the oracle standing in for the per-rank GPU compute
(tests may use the oracle), against the oracle's transform of the whole vector (ntt.rs:7-64).  All ranks inside one process
(exchanges as slicing) for W = 1..8, and world size 2 over gloo with the real all_to_all_single."""
import os, socket, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import orc
from myzkp_amd import sharded

LAYOUTS = [("contiguous", "contiguous"), ("contiguous", "cyclic"), ("cyclic", "contiguous")]


class OracleOps:
    """The local steps by the oracle: numpy (elements, limbs) uint64 arrays."""

    def __init__(self, fid):
        self.fid, self.nl, self.p = fid, orc.LIMBS[fid], orc.MOD[fid]

    def ntt(self, b, m, root, inverse):
        assert b.shape[0] == m
        rc, out = orc.ntt_fast(self.fid, root, np.ascontiguousarray(b), inverse, 1)
        assert rc == 0
        return out

    def ntt_rows(self, b, rows, n, root, inverse):
        return np.concatenate([self.ntt(b[i * n:(i + 1) * n], n, root, inverse) for i in range(rows)])

    def lde(self, b, m, offset, generator):
        rc, out = orc.coset_ref(self.fid, np.ascontiguousarray(b), offset, generator, m)
        assert rc == 0
        return out

    def scale(self, b, m, ratio):
        v = orc.from_limbs(b)
        return orc.to_limbs([x * pow(ratio, i, self.p) % self.p for i, x in enumerate(v)], self.nl)

    def transpose(self, b, rows, cols):
        return np.ascontiguousarray(b.reshape(rows, cols, self.nl).transpose(1, 0, 2)).reshape(-1, self.nl)

    def chunks(self, b, W):
        return np.split(b, W)

    def cat(self, parts):
        return np.concatenate(parts)

    def all_to_all(self, b, group=None):
        src = torch.from_numpy(np.ascontiguousarray(b).view(np.int64).reshape(-1).copy())
        out = torch.empty_like(src)
        dist.all_to_all_single(out, src, group=group)
        return out.numpy().view(np.uint64).reshape(-1, self.nl)


def _parts(x, W, layout):
    m = x.shape[0] // W
    return [x[r * m:(r + 1) * m] for r in range(W)] if layout == "contiguous" else [x[r::W] for r in range(W)]


@pytest.mark.parametrize("fid", [orc.FR, orc.M128])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("lin,lout", LAYOUTS)
def test_sharded_schedule_all_ranks_in_process(fid, inverse, lin, lout):
    ops = OracleOps(fid)
    for lg, worlds in ((6, (1, 2, 4, 8)), (9, (2, 8))):
        n = 1 << lg
        w = orc.root_of(fid, lg)
        x = orc.synth_vector(fid, 600 + lg, n, 1)
        rc, want = orc.ntt_fast(fid, w, x, inverse, 1)
        assert rc == 0
        for W in worlds:
            outs = sharded.ntt_sharded_simulate(_parts(x, W, lin), orc.MOD[fid], lg, w, ops, inverse, lin, lout)
            for r, (got, exp) in enumerate(zip(outs, _parts(want, W, lout))):
                assert np.array_equal(got, exp), (lg, W, r)


def test_sharded_schedule_rejects_bad_shapes():
    with pytest.raises(ValueError):
        sharded.ntt_sharded_steps(orc.MOD[orc.FR], 4, 8, 5, False, "contiguous", "contiguous")     # world^2 > n
    with pytest.raises(ValueError):
        sharded.ntt_sharded_steps(orc.MOD[orc.FR], 8, 3, 5, False, "contiguous", "contiguous")     # world not a power of two
    with pytest.raises(ValueError):
        sharded.ntt_sharded_steps(orc.MOD[orc.FR], 8, 2, 5, False, "cyclic", "cyclic")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for fid in (orc.FR, orc.M128):
        lg = 8
        w = orc.root_of(fid, lg)
        x = orc.synth_vector(fid, 77, 1 << lg, 1)
        ops = OracleOps(fid)
        for inverse in (False, True):
            rc, want = orc.ntt_fast(fid, w, x, inverse, 1)
            for lin, lout in LAYOUTS:
                got = sharded.ntt_sharded(_parts(x, world, lin)[rank], orc.MOD[fid], lg, w, ops, inverse, lin, lout)
                ok = ok and rc == 0 and np.array_equal(got, _parts(want, world, lout)[rank])
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ntt_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
