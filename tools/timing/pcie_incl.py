"""The host-buffer entry points with their transfers inside (pageable host memory): python tools/timing/pcie_incl.py [msm|commit|ntt ...]
(no argument: all).  Results are checked against the first call's."""
import sys, os, time, ctypes
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import myzkp_amd as mz, orc
mz.init(0)
which = set(sys.argv[1:]) or {"msm", "commit", "ntt"}
n = 1 << 20
s = orc.synth_vector(orc.FR, 1, n, 64); p = orc.synth_points(2, n, 64)
L = mz.lib()


def clock(name, fn, unit="pairs"):
    first = fn(); fn()
    t0 = time.perf_counter()
    for _ in range(8):
        r = fn()
    dt = (time.perf_counter() - t0) / 8
    same = "" if first is None else ("  same result every call: %s" % (r == first))
    print("%s: %.3f ms per call, %.3g %s/s%s" % (name, dt * 1e3, n / dt, unit, same), flush=True)


if "msm" in which:
    clock("msm host-buffer (generic)", lambda: mz.msm_g1(s, p))
if "ntt" in which:
    v = orc.synth_vector(orc.FR, 3, n, 64); w = orc.fr_root(20)
    vout = np.zeros_like(v)          # the caller's output vector, allocated (and touched) once: a fresh array per call costs page faults
    wl = mz.to_limbs([w], 4)

    def ntt_c():
        assert L.mzk_ntt(0, wl.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), vout.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), 0) == 0
    clock("ntt host-buffer, caller's output vector reused (C ABI)", ntt_c, "elems")
if "commit" in which:
    h = mz.Srs(p)
    clock("kzg commit host scalars + resident SRS", lambda: h.commit(s))
    h.close()
