import sys, time, ctypes; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, myzkp_amd as mz
mz.init(0); L = mz.lib(); dev = torch.device("cuda", 0)
L.mzk_prof_name.restype = ctypes.c_char_p
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in [int(x) for x in sys.argv[1].split(",")]:
    n = 1 << lg
    pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    assert L.mzk_synth_g1_points_dev(ctypes.c_uint64(7), ctypes.c_size_t(n), ctypes.c_void_p(pts.data_ptr()), st) == 0
    assert L.mzk_synth_field_dev(0, ctypes.c_uint64(9), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st) == 0
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    ref = None
    for c in [int(x) for x in sys.argv[2].split(",")]:
        h = ctypes.c_void_p()
        t0 = time.perf_counter()
        assert L.mzk_srs_from_device_ex(ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.c_int(c), ctypes.byref(h), st) == 0, L.mzk_last_error()
        torch.cuda.synchronize(); tb = time.perf_counter() - t0
        def f():
            assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st) == 0, L.mzk_last_error()
        for _ in range(3): f()
        torch.cuda.synchronize()
        L.mzk_prof_reset(); L.mzk_prof_enable(1)
        t0 = time.perf_counter(); K = 5
        for _ in range(K): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        L.mzk_prof_enable(0)
        ph = {}
        for p in range(13):
            ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
            L.mzk_prof_read(p, ctypes.byref(ms), ctypes.byref(cnt))
            if cnt.value: ph[L.mzk_prof_name(p).decode()[4:]] = round(ms.value / cnt.value, 3)
        res = out.cpu().numpy().tobytes()
        if ref is None: ref = res
        print(f"2^{lg} c={c}: {dt*1e3:.2f} ms  tables {tb*1e3:.0f} ms  same_result={res == ref}  {ph}", flush=True)
        L.mzk_srs_free(h); torch.cuda.empty_cache()
