"""Merkle commits of 2^3 .. 2^12 M128 leaves (device-resident): the differences between consecutive sizes are what one more level of the
single-workgroup tail costs:  python tools/timing/merkle_small_levels.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
root = (ctypes.c_uint8 * 48)(); ln = ctypes.c_size_t()
prev = None
for lg in range(3, 13):
    n = 1 << lg
    d = torch.zeros(n * 2, dtype=torch.int64, device=dev)
    assert L.mzk_synth_field_dev(1, ctypes.c_uint64(5), ctypes.c_size_t(n), ctypes.c_void_p(d.data_ptr()), st) == 0
    f = lambda: L.mzk_merkle_commit_field_dev(1, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), root, ctypes.c_size_t(48), ctypes.byref(ln), st)
    for _ in range(20): assert f() == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 200 * 1e6
    print("2^%-2d leaves: %7.1f us per commit%s" % (lg, us, "" if prev is None else "   (+%.1f us over 2^%d)" % (us - prev, lg - 1)), flush=True)
    prev = us
