"""Per-rank LOCAL cost of one sharded transform (myzkp_amd/sharded.py): all W ranks' steps run one after the other on one GPU
with the exchanges as slicing, so (total - slicing) / W is what each GPU would compute between its all-to-alls.  The exchange
itself is not measured here (one GPU): bytes per rank and exchange are printed instead."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import myzkp_amd as mz
from myzkp_amd import sharded
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fid, nl = mz.FIELD_FR, 4
p = mz.MODULUS[fid]
ops = sharded.DeviceOps(fid)


class Timed:
    """wraps the ops: accumulates GPU time spent inside the local steps (events), leaves chunks/cat (the fake exchange) out"""
    def __init__(self, ops):
        self.ops, self.ms = ops, 0.0
    def __getattr__(self, name):
        f = getattr(self.ops, name)
        if name in ("chunks", "cat"):
            return f
        def g(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = f(*a, **k); e1.record(); e1.synchronize()
            self.ms += e0.elapsed_time(e1)
            return r
        return g


for lg in [int(a) for a in sys.argv[1:]] or [24]:
    n = 1 << lg
    w = mz.root_of_unity(fid, lg)
    x = torch.empty(n * nl, dtype=torch.int64, device=dev)
    assert L.mzk_synth_field_dev(fid, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(x.data_ptr()), st) == 0
    y = torch.empty_like(x)
    r = mz.to_limbs([w], nl)
    def single():
        assert L.mzk_ntt_dev(fid, r.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_size_t(n), 0, st) == 0
    single(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): single()
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 5 * 1e3
    print("Fr 2^%d on one GPU: %.3f ms" % (lg, t1), flush=True)
    for W in (2, 4, 8):
        m = n // W
        for lin, lout in (("contiguous", "cyclic"), ("contiguous", "contiguous"), ("cyclic", "contiguous")):
            parts = [x.view(W, -1)[i] for i in range(W)] if lin == "contiguous" else [x.view(-1, nl)[i::W].contiguous().view(-1) for i in range(W)]
            sharded.ntt_sharded_simulate(parts, p, lg, w, ops, False, lin, lout)          # plans, allocator
            t = Timed(ops)
            sharded.ntt_sharded_simulate(parts, p, lg, w, t, False, lin, lout)
            nx = 2 if "cyclic" in (lin, lout) else 3
            print("  W=%d %-10s -> %-10s local steps %.3f ms per rank; %d exchanges of %.1f MiB sent per rank"
                  % (W, lin, lout, t.ms / W, nx, (W - 1) * (m // W) * 32 / 2**20), flush=True)
