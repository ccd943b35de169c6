import sys, time, ctypes; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, myzkp_amd as mz
mz.init(0); L = mz.lib(); dev = torch.device("cuda", 0)
L.mzk_prof_name.restype = ctypes.c_char_p
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,10,12,14,16,18,19").split(",")]:
    n = 1 << lg
    pts = torch.empty(n * 8, dtype=torch.int64, device=dev); sc = torch.empty(n * 4, dtype=torch.int64, device=dev); out = torch.zeros(8, dtype=torch.int64, device=dev)
    assert L.mzk_synth_g1_points_dev(ctypes.c_uint64(7), ctypes.c_size_t(n), ctypes.c_void_p(pts.data_ptr()), st) == 0
    assert L.mzk_synth_field_dev(0, ctypes.c_uint64(9), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st) == 0
    def f():
        assert L.mzk_msm_g1_bn254_dev(ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), st) == 0
    for _ in range(3): f()
    torch.cuda.synchronize(); L.mzk_prof_reset(); L.mzk_prof_enable(1)
    t0 = time.perf_counter(); K = 10
    for _ in range(K): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    L.mzk_prof_enable(0)
    ph = {}
    for p in range(11):
        ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
        L.mzk_prof_read(p, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value: ph[L.mzk_prof_name(p).decode()[4:]] = round(ms.value / cnt.value, 3)
    print(f"generic 2^{lg}: {dt*1e3:.3f} ms {ph}", flush=True)
