"""Batched NTT (mzk_ntt_batch_dev): time per transform for batches of 1 .. 64 columns, 2^10 .. 2^20 points, BN254 Fr.
python tools/timing/ntt_batch.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nl = 4 if fid == 0 else 2
for lg in (10, 12, 14, 16, 18, 20):
    n = 1 << lg
    root = mz.to_limbs([mz.root_of_unity(fid, lg)], nl)
    row = []
    for batch in ([int(x) for x in os.environ["NTT_BATCHES"].split(",")] if os.environ.get("NTT_BATCHES") else (1, 4, 16, 64)):
        if batch * n > (1 << 24):
            continue
        v = torch.empty(batch * n * nl, dtype=torch.int64, device=dev)
        for k in range(batch):
            L.mzk_synth_field_dev(fid, ctypes.c_uint64(5 + k), ctypes.c_size_t(n), ctypes.c_void_p(v.data_ptr() + k * n * nl * 8), st)
        o = torch.empty_like(v)
        def run():
            assert L.mzk_ntt_batch_dev(fid, root.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(v.data_ptr()), ctypes.c_void_p(o.data_ptr()),
                                       ctypes.c_size_t(n), ctypes.c_size_t(batch), 0, st) == 0, L.mzk_last_error()
        run(); run(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row.append("batch %2d: %.4f ms (%.4f per transform, %.2e elem/s)" % (batch, ms, ms / batch, batch * n / ms * 1e3))
    print("%s 2^%d  " % ("Fr" if fid == 0 else "M128", lg) + " | ".join(row), flush=True)
