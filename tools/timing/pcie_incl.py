import sys, os, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import myzkp_amd as mz, orc
mz.init(0)
n = 1 << 20
s = orc.synth_vector(orc.FR, 1, n, 64); p = orc.synth_points(2, n, 64)
v = orc.synth_vector(orc.FR, 3, n, 64); w = orc.fr_root(20)
import ctypes
L = mz.lib()
vout = np.zeros_like(v)          # the caller's output vector, allocated (and touched) once: a fresh array per call costs page faults
wl = mz.to_limbs([w], 4)
def ntt_c():
    assert L.mzk_ntt(0, wl.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), vout.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), 0) == 0
for name, fn in (("msm host-buffer (generic)", lambda: mz.msm_g1(s, p)), ("ntt host-buffer, fresh output array per call (python wrapper)", lambda: mz.ntt(orc.FR, w, v)),
                 ("ntt host-buffer, caller's output vector reused (C ABI)", ntt_c)):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    print("%s: %.2f ms per call, %.3g units/s" % (name, dt * 1e3, n / dt))
h = mz.Srs(p)
h.commit(s); t0 = time.perf_counter()
for _ in range(5): h.commit(s)
dt = (time.perf_counter() - t0) / 5
print("kzg commit host scalars + resident SRS: %.2f ms per call, %.3g pairs/s" % (dt * 1e3, n / dt))
