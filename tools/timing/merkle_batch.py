"""Batched Merkle commits (mzk_merkle_commit_field_batch_dev): ms per call for 1 / 4 / 16 / 64 codewords of 2^12 .. 2^20 M128
elements.   python tools/timing/merkle_batch.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in (12, 14, 16, 18, 20):
    n = 1 << lg
    row = []
    for batch in (1, 4, 16, 64):
        if batch * n > (1 << 24):
            continue
        d = torch.empty(batch * n * 2, dtype=torch.int64, device=dev)
        for k in range(batch):
            L.mzk_synth_field_dev(1, ctypes.c_uint64(5 + k), ctypes.c_size_t(n), ctypes.c_void_p(d.data_ptr() + k * n * 16), st)
        roots = (ctypes.c_uint8 * (32 * batch))()
        def run():
            assert L.mzk_merkle_commit_field_batch_dev(1, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(batch), roots, st) == 0, L.mzk_last_error()
        for _ in range(3): run()
        t0 = time.perf_counter(); reps = 20
        for _ in range(reps): run()
        ms = (time.perf_counter() - t0) / reps * 1e3
        row.append("batch %2d: %.3f ms (%.3f per tree)" % (batch, ms, ms / batch))
    print("M128 2^%d leaves  " % lg + " | ".join(row), flush=True)
