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
# the contexts' own streams (distinct priorities = distinct hardware queues); PIPE_TORCH_STREAMS=1 uses two torch
# streams instead, which overlap only when the runtime happens to put them on different queues
mz.lib().mzk_ctx_stream.restype = ctypes.c_void_p
st = [ctypes.c_void_p(mz.lib().mzk_ctx_stream(k)) for k in range(2)]
if os.environ.get("PIPE_TORCH_STREAMS") == "1":
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
if len(sys.argv) > 2 and sys.argv[2] == "--after-work":
    # does other work issued on the (legacy) default stream before change the overlap?  (bench.py runs this leg last)
    v = torch.empty(n * 4, dtype=torch.int64, device=dev)
    root = mz.to_limbs([mz.root_of_unity(0, lg)], 4)
    for _ in range(200):
        L.mzk_ntt_dev(0, root.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(sc[0].data_ptr()), ctypes.c_void_p(v.data_ptr()), ctypes.c_size_t(n), 0, d0)
        L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc[0].data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, d0)
    torch.cuda.synchronize()
    for two in (False, True):
        run(two, 40)
        ms = min(run(two, 100) for _ in range(3))
        print("after default-stream work: %s: %.3f ms per commit" % ("two contexts alternating" if two else "one context", ms), flush=True)
    # same with the library's profiling toggled the way bench.py does
    L.mzk_prof_reset(); L.mzk_prof_select(ctypes.c_uint32(4)); L.mzk_prof_enable(1); run(False, 10); L.mzk_prof_enable(0); L.mzk_prof_reset()
    for two in (False, True):
        run(two, 40)
        print("after a profiled pass: %s: %.3f ms per commit" % ("two" if two else "one", min(run(two, 100) for _ in range(3))), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "--bisect":
    def gain(label):
        a = min(run(False, 100) for _ in range(2)); b = min(run(True, 100) for _ in range(2))
        print("%-50s one %.3f  two %.3f  (%.1f %%)" % (label, a, b, (a / b - 1) * 100), flush=True)
    gain("baseline")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
    import orc, numpy as np
    threads = os.cpu_count() or 1
    s_cpu = orc.synth_vector(orc.FR, 5, 1 << 18, threads); p_cpu = orc.synth_points(6, 1 << 14, threads)
    gain("after oracle OpenMP work (%d threads)" % threads)
    orc.msm_fast(s_cpu[:1 << 14], p_cpu, threads)
    gain("after oracle msm_fast")
    x = sc[0].cpu().numpy()
    gain("after a D2H copy through torch")
    import torch.distributed as dist
    gain("after importing torch.distributed")
    mroot, mlen = (ctypes.c_uint8 * 48)(), ctypes.c_size_t()
    m_in = torch.empty(n * 2, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(1, ctypes.c_uint64(9), ctypes.c_size_t(n), ctypes.c_void_p(m_in.data_ptr()), d0)
    for _ in range(20):
        L.mzk_merkle_commit_field_dev(1, ctypes.c_void_p(m_in.data_ptr()), ctypes.c_size_t(n), mroot, ctypes.c_size_t(48), ctypes.byref(mlen), d0)
    gain("after Merkle commits (D2H of the root per call)")
    a = torch.empty(1 << 28, dtype=torch.int32, device=dev); b2 = torch.empty_like(a); b2.copy_(a); torch.cuda.synchronize(); del a, b2
    gain("after a 1 GiB torch copy")
