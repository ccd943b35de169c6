"""The open stage of the end-to-end KZG leg taken apart (bench.py leg_e2e_kzg, world 1): quotient pass, the MSM of the quotient, the MSM of the
coefficients, each bracketed by a device synchronize; then the stage as the bench runs it.
    python tools/timing/e2e_open_split.py [lg = 22]"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, orc, myzkp_amd as mz
from myzkp_amd import sharded
mz.init(0)
L = mz.lib()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
dev = "cuda"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
cf = torch.empty(n * 4, dtype=torch.int64, device=dev)
sp = torch.empty(n * 8, dtype=torch.int64, device=dev)
assert L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(5), ctypes.c_size_t(n), p(cf), st) == 0
alpha = orc.from_limbs(orc.synth_vector(orc.FR, 6, 1))[0]
uu = orc.from_limbs(orc.synth_vector(orc.FR, 7, 1))[0]
a_l, g_l = mz.to_limbs([alpha], 4), mz.points_to_array([(1, 2)])
assert L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(0), ctypes.c_size_t(n), p(sp), st) == 0
hh = ctypes.c_void_p()
assert L.mzk_srs_from_device_ex(p(sp), ctypes.c_size_t(n), ctypes.c_int(0), ctypes.byref(hh), st) == 0
rec = torch.zeros(16, dtype=torch.int64, device=dev)
q_buf = torch.zeros(n * 4, dtype=torch.int64, device=dev)
ops = sharded.DeviceOpenOps(out=q_buf)


def timed(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def msm(t):
    assert L.mzk_kzg_commit_srs_dev(hh, p(t), ctypes.c_size_t(n), p(rec), ctypes.c_int(1), st) == 0


state = {}


def quotient():
    state["y"], state["q"] = sharded.sharded_open_quotient(ops, cf, n, uu, orc.P_FR, 0, 1)


def stage():
    quotient(); msm(state["q"])


print("MSM of the coefficients:      %.3f ms" % timed(lambda: msm(cf)))
print("quotient pass (host gets y):  %.3f ms" % timed(quotient))
print("MSM of the quotient:          %.3f ms" % timed(lambda: msm(state["q"])))
print("open stage as the bench runs: %.3f ms" % timed(stage))
print("MSM of the coefficients:      %.3f ms" % timed(lambda: msm(cf)))
qq = state["q"].cpu().numpy().view(np.uint64).reshape(n, 4)
print("quotient: last element zero: %s; elements with a zero top limb: %d of %d" % (not qq[-1].any(), int((qq[:, 3] == 0).sum()), n))
