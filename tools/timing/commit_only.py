"""Only KZG commits against an SRS handle with the default window width (for rocprofv3 --pmc / --kernel-trace runs that
must not mix in the generic layout's launches of the same kernels): python tools/timing/commit_only.py [LOG=20] [REPS=40] [WINDOW_BITS=1]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import myzkp_amd as mz

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
wb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0); n = 1 << lg
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
sc = torch.empty(n * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
h = ctypes.c_void_p()
assert L.mzk_srs_from_device_ex(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), wb, ctypes.byref(h), st) == 0
out = torch.zeros(16, dtype=torch.int64, device=dev)
for _ in range(5):
    assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st) == 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st)
torch.cuda.synchronize()
print("commit 2^%d, window option %d: %.3f ms per commit" % (lg, wb, (time.perf_counter() - t0) / reps * 1e3))
