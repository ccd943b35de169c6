"""Grid-batched openings (mzk_kzg_open_srs_many_dev: open_kzg per polynomial at its own point, das/avail.rs:132) against the same
openings one call at a time: python tools/timing/many_open.py [lg_n:count[:direct],...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import myzkp_amd as mz
import orc
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for spec in (sys.argv[1] if len(sys.argv) > 1 else "10:256,10:256:10,12:64,8:1024").split(','):
    f = [int(x) for x in spec.split(':')]
    lg, count, direct = f[0], f[1], (f[2] if len(f) > 2 else 0)
    n = 1 << lg
    sc = torch.empty(n * count * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n * count), ctypes.c_void_p(sc.data_ptr()), st)
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
    if direct:
        assert L.mzk_srs_build_direct(h, direct, ctypes.c_size_t(64 << 30), st) == 0, L.mzk_last_error()
    us = orc.synth_vector(orc.FR, 77, count)
    ys = torch.zeros(count * 4, dtype=torch.int64, device=dev); ws = torch.zeros(count * 8, dtype=torch.int64, device=dev)
    ys1 = torch.zeros(count * 4, dtype=torch.int64, device=dev); ws1 = torch.zeros(count * 8, dtype=torch.int64, device=dev)
    def many():
        return L.mzk_kzg_open_srs_many_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), us.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_void_p(ys.data_ptr()), ctypes.c_void_p(ws.data_ptr()), st)
    def loop():
        for k in range(count):
            rc = L.mzk_kzg_open_srs_dev(h, ctypes.c_void_p(sc.data_ptr() + k * n * 32), ctypes.c_size_t(n), us[k].ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_void_p(ys1.data_ptr() + k * 32), ctypes.c_void_p(ws1.data_ptr() + k * 64), st)
            if rc: return rc
        return 0
    res = []
    for fn, reps in ((many, 20), (loop, 2)):
        for _ in range(2):
            assert fn() == 0, L.mzk_last_error()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e3)
    print("%4d x 2^%-2d%s: many %.3f ms = %.2f us per opening | one at a time %.3f ms = %.1f us per opening | %.1fx | same: %s"
          % (count, lg, " (+%d-bit direct tables)" % direct if direct else "", res[0], res[0] / count * 1e3, res[1], res[1] / count * 1e3, res[1] / res[0],
             bool(torch.equal(ys, ys1) and torch.equal(ws, ws1))), flush=True)
    L.mzk_srs_free(h)
