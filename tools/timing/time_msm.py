import ctypes, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0); n = 1 << 20
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
sc = torch.empty(n*4, dtype=torch.int64, device=dev); pt = torch.empty(n*8, dtype=torch.int64, device=dev)
L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
out = torch.zeros(16, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for it in range(6):
    t0 = time.perf_counter()
    rc = L.mzk_msm_g1_bn254_partial_dev(ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("iter", it, "rc", rc, "enqueue ms %.3f total ms %.3f" % ((t1-t0)*1e3, (t2-t0)*1e3))
