import sys, time, ctypes; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, myzkp_amd as mz
mz.init(0); L = mz.lib(); dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in (20, 22, 24):
    n = 1 << lg
    cf = torch.empty(n * 4, dtype=torch.int64, device=dev); q = torch.empty(n * 4, dtype=torch.int64, device=dev); y = torch.zeros(4, dtype=torch.int64, device=dev)
    assert L.mzk_synth_field_dev(0, ctypes.c_uint64(9), ctypes.c_size_t(n), ctypes.c_void_p(cf.data_ptr()), st) == 0
    u = mz.to_limbs([123456789123456789], 4)
    def f():
        assert L.mzk_kzg_open_quotient_dev(ctypes.c_void_p(cf.data_ptr()), ctypes.c_size_t(n), u.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(q.data_ptr()), st) == 0
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print("open quotient 2^%d: %.3f ms" % (lg, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
