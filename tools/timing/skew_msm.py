import sys, time, ctypes; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, orc, myzkp_amd as mz
mz.init(0); L = mz.lib(); dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 1 << 20
pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
assert L.mzk_synth_g1_points_dev(ctypes.c_uint64(7), ctypes.c_size_t(n), ctypes.c_void_p(pts.data_ptr()), st) == 0
h = ctypes.c_void_p()
assert L.mzk_srs_from_device(ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
out = torch.zeros(8, dtype=torch.int64, device=dev)
def run(sc, label):
    d = torch.from_numpy(sc.view(np.int64).reshape(-1).copy()).to(dev)
    for kind in ("generic", "merged"):
        def f():
            if kind == "generic":
                assert L.mzk_msm_g1_bn254_dev(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), st) == 0
            else:
                assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st) == 0
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize(); print(label, kind, "%.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
uni = orc.synth_vector(orc.FR, 3, n)
run(uni, "uniform")
eq = np.tile(uni[:1], (n, 1)); run(eq, "all-equal")
two = uni.copy(); two[:, 1:] = 0; two[:, 0] &= np.uint64(0xffff); run(two, "16-bit scalars")
one = np.zeros((n, 4), dtype=np.uint64); one[:, 0] = 1; run(one, "all ones")
