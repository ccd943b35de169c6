"""Cost of the final XYZZ -> affine conversion (one safegcd inversion on one lane) and of a fold of W partial records:
mzk_g1_fold_partials_dev with W = 1, 2, 8 records, one call at a time (kernel durations under rocprofv3 --kernel-trace)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 4096
sc = torch.empty(n * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
recs = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
for r in range(8):
    assert L.mzk_msm_g1_bn254_partial_dev(ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(pt.data_ptr() + 64 * 100 * r), ctypes.c_size_t(500), ctypes.c_void_p(recs.data_ptr() + 128 * r), st) == 0
out = torch.zeros(8, dtype=torch.int64, device=dev)
for W in (1, 2, 8):
    for _ in range(5):
        L.mzk_g1_fold_partials_dev(ctypes.c_void_p(recs.data_ptr()), ctypes.c_int(W), ctypes.c_void_p(out.data_ptr()), st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        L.mzk_g1_fold_partials_dev(ctypes.c_void_p(recs.data_ptr()), ctypes.c_int(W), ctypes.c_void_p(out.data_ptr()), st); torch.cuda.synchronize()
    print("fold of %d partial(s): %.1f us per call (host clock, synchronised)" % (W, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
