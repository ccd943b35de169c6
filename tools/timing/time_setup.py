import sys, time, ctypes; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, myzkp_amd as mz
mz.init(0); L = mz.lib(); dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
a_l, g_l = mz.to_limbs([0x123456789abcdef], 4), mz.points_to_array([(1, 2)])
for lg in (10, 14, 16, 20, 22):
    n = 1 << lg
    out = torch.empty(n * 8, dtype=torch.int64, device=dev)
    def f():
        assert L.mzk_kzg_setup_g1_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n - 1), ctypes.c_void_p(out.data_ptr()), st) == 0
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); first = time.perf_counter() - t0
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); print("setup 2^%d: first %.2f ms, steady %.3f ms" % (lg, first * 1e3, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
