"""Where the window-table build of a large SRS spends its time: allocation vs kernels (one-off cost per SRS)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
hip = ctypes.CDLL("libamdhip64.so")
def sync(): torch.cuda.synchronize()
for gib in (1, 16, 64):
    p = ctypes.c_void_p(); sync(); t0 = time.perf_counter()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gib << 30)) == 0
    t1 = time.perf_counter(); hip.hipFree(p); t2 = time.perf_counter()
    print("hipMalloc %3d GiB: %.1f ms, hipFree %.1f ms" % (gib, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
a_l, g_l = mz.to_limbs([12345], 4), mz.points_to_array([(1, 2)])
assert L.mzk_kzg_setup_g1_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n - 1), ctypes.c_void_p(pts.data_ptr()), st) == 0
sync()
L.mzk_prof_enable(1)
for rep in range(3):
    h = ctypes.c_void_p(); t0 = time.perf_counter()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
    sync(); t1 = time.perf_counter()
    L.mzk_srs_free(h); t2 = time.perf_counter()
    print("2^%d window tables, build %d: %.1f ms (free %.1f ms)" % (lg, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
