"""Latency of small MSMs / KZG commits (the reference's real sizes): generic MSM on arbitrary points vs commit against
an SRS handle, 2^4 .. 2^13 pairs, device-resident inputs, one call at a time.   python tools/timing/small_latency.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (4, 8, 10, 11, 12, 13, 14)):
    n = 1 << lg
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    res = []
    for fn in (lambda: L.mzk_msm_g1_bn254_dev(ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), st),
               lambda: L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr() + 64), 0, st)):
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            assert fn() == 0; torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(); torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e3)
    same = bool(torch.equal(out[:8], out[8:]))
    print("2^%-2d pairs: generic MSM %.3f ms   commit vs SRS handle %.3f ms   same point: %s" % (lg, res[0], res[1], same), flush=True)
    L.mzk_srs_free(h)
