"""2^20-pair MSM / KZG commit on skewed and SHORT scalars (witness-like inputs: bits, bytes, 16/32/64-bit values, repeated values), both
layouts, with the per-phase split of the profiler:   python tools/timing/skew_msm.py [pattern,...]"""
import sys, time, ctypes, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, orc, myzkp_amd as mz
mz.init(0); L = mz.lib(); dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 1 << 20
pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
assert L.mzk_synth_g1_points_dev(ctypes.c_uint64(7), ctypes.c_size_t(n), ctypes.c_void_p(pts.data_ptr()), st) == 0
h = ctypes.c_void_p()
assert L.mzk_srs_from_device(ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
out = torch.zeros(8, dtype=torch.int64, device=dev)
PH = {}
L.mzk_prof_name.restype = ctypes.c_char_p
for i in range(32):
    nm = L.mzk_prof_name(i)
    if nm: PH[i] = nm.decode()
def run(sc, label):
    d = torch.from_numpy(sc.view(np.int64).reshape(-1).copy()).to(dev)
    for kind in ("generic", "merged"):
        def f():
            if kind == "generic":
                assert L.mzk_msm_g1_bn254_dev(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), st) == 0
            else:
                assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st) == 0
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        L.mzk_prof_reset(); L.mzk_prof_enable(1)
        for _ in range(5): f()
        torch.cuda.synchronize()
        split = []
        for i, nm in PH.items():
            v = ctypes.c_double(); cnt = ctypes.c_uint64()
            L.mzk_prof_read(i, ctypes.byref(v), ctypes.byref(cnt))
            if cnt.value: split.append("%s %.3f" % (nm.replace("msm_", ""), v.value / 5))
        L.mzk_prof_enable(0)
        print("%-16s %-8s %.2f ms | %s" % (label, kind, ms, "  ".join(split)), flush=True)
uni = orc.synth_vector(orc.FR, 3, n)
def low(bits):
    v = np.zeros((n, 4), dtype=np.uint64)
    full, rem = bits // 64, bits % 64
    v[:, :full] = uni[:, :full]
    if rem: v[:, full] = uni[:, full] & np.uint64((1 << rem) - 1)
    return v
one = np.zeros((n, 4), dtype=np.uint64); one[:, 0] = 1
half = uni.copy(); half[::2] = 0
pats = {"uniform": uni, "all-equal": np.tile(uni[:1], (n, 1)), "all ones": one, "bits": low(1), "bytes": low(8), "16-bit scalars": low(16), "32-bit": low(32),
        "64-bit": low(64), "128-bit": low(128), "248-bit": low(248), "half zero": half}
want = sys.argv[1].split(",") if len(sys.argv) > 1 else ["uniform", "all-equal", "16-bit scalars", "all ones"]
for k in want:
    run(pats[k], k)
