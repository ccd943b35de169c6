"""The quotient pass of open_kzg alone (mzk_kzg_open_quotient_dev: y = f(u) and q = (f - y) / (X - u), kzg.rs:61-69) at 2^lg coefficients, device-resident,
beside the generic MSM of the same size it feeds:
    python tools/timing/open_quotient_time.py [lg = 22]"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, orc, myzkp_amd as mz
mz.init(0)
L = mz.lib()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
coef = torch.empty(n * 4, dtype=torch.int64, device="cuda")
assert L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(77), ctypes.c_size_t(n), ctypes.c_void_p(coef.data_ptr()), st) == 0
q = torch.empty_like(coef)
y = torch.zeros(4, dtype=torch.int64, device="cuda")
u = mz.to_limbs([0x1234567890abcdef1234567890abcdef % orc.P_FR], 4)


def run():
    assert L.mzk_kzg_open_quotient_dev(ctypes.c_void_p(coef.data_ptr()), ctypes.c_size_t(n), u.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(y.data_ptr()),
                                       ctypes.c_void_p(q.data_ptr()), st) == 0


for _ in range(3):
    run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
print("open quotient 2^%d: %.3f ms per call" % (lg, (time.perf_counter() - t0) / 10 * 1e3))
# parity of the value on a sample: y = f(u) by the oracle's Horner on the host copy
cf = coef.cpu().numpy().view(np.uint64).reshape(n, 4)
if lg <= 22:
    want = orc.poly_eval(orc.FR, cf, orc.from_limbs(u)[0])
    print("y equals the oracle's Horner value:", orc.from_limbs(y.cpu().numpy().view(np.uint64).reshape(1, 4))[0] == want)
