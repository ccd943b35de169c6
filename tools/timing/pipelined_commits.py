"""Throughput of back-to-back KZG commits against one SRS when TWO contexts on the same GPU alternate (each with its own
stream and workspace): the latency-bound tail of commit i (bucket reduction, inversion: ~0.3 ms on a nearly idle GPU)
overlaps the sort / accumulate of commit i + 1.   python tools/timing/pipelined_commits.py [log2n]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
mz.init_devices([0, 0]); L = mz.lib()
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
st = [ctypes.c_void_p(s.cuda_stream) for s in streams]
sc = [torch.empty(n * 4, dtype=torch.int64, device=dev) for _ in range(2)]
pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
d0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for k in range(2):
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1 + k), ctypes.c_size_t(n), ctypes.c_void_p(sc[k].data_ptr()), d0)
L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), d0)
h = ctypes.c_void_p()
assert L.mzk_srs_from_device(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), d0) == 0
torch.cuda.synchronize()
out = torch.zeros(4 * 8, dtype=torch.int64, device=dev)
def commit(ctx, k, slot):
    mz.ctx_select(ctx)
    assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc[k].data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr() + 64 * slot), 0, st[ctx]) == 0, L.mzk_last_error()
def run(two, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        commit(i & 1 if two else 0, i & 1, (i & 1) + (2 if two else 0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for two in (False, True):
    run(two, 40)
    ms = min(run(two, 100) for _ in range(3))
    print("2^%d pairs, %s: %.3f ms per commit = %.3g pairs/s" % (lg, "two contexts alternating" if two else "one context", ms, n / ms * 1e3), flush=True)
o = out.cpu().numpy().reshape(4, 8)
print("results identical:", bool((o[0] == o[2]).all() and (o[1] == o[3]).all()))
mz.ctx_select(0)
