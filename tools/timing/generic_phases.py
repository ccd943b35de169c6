"""Phases of the generic G1 MSM (arbitrary points, GLV layout) from the library's profiler: python tools/timing/generic_phases.py [log2n ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
L.mzk_prof_name.restype = ctypes.c_char_p
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in ([int(a) for a in sys.argv[1:]] or [20]):
    n = 1 << lg
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    def f():
        assert L.mzk_msm_g1_bn254_partial_dev(ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), st) == 0
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 20
    for _ in range(K): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    L.mzk_prof_reset(); L.mzk_prof_enable(1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    L.mzk_prof_enable(0)
    ph = {}
    for p in range(13):
        ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
        L.mzk_prof_read(p, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value: ph[L.mzk_prof_name(p).decode()[4:]] = round(ms.value / cnt.value, 3)
    print("generic MSM 2^%d: %.3f ms per call  %s" % (lg, dt * 1e3, ph), flush=True)
