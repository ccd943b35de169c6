import ctypes, os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for fid, nl, name in ((1, 2, "M128"), (0, 4, "Fr")):
    for lg in (16, 20, 24):
        n = 1 << lg
        d = torch.zeros(n * nl, dtype=torch.int64, device=dev)
        assert L.mzk_synth_field_dev(fid, ctypes.c_uint64(5), ctypes.c_size_t(n), ctypes.c_void_p(d.data_ptr()), st) == 0
        root = (ctypes.c_uint8 * 48)(); ln = ctypes.c_size_t()
        def run():
            assert L.mzk_merkle_commit_field_dev(fid, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), root, ctypes.c_size_t(48), ctypes.byref(ln), st) == 0
        for _ in range(5): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 20
        for _ in range(K): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print(f"{name} 2^{lg}: {dt*1e3:.3f} ms  {n/dt/1e9:.2f} G leaves/s  {(n-1)/dt/1e9:.2f} G hashes/s")
