"""Phase timestamps of k_reduce_tail_row from a -DMZK_TAIL_TRACE build of the library (MZK_HIP_LIB selects it):
python tools/timing/tail_trace.py [log2n ...]   -- commits against an SRS handle, 100-MHz device clock, last of three runs."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import myzkp_amd as mz

mz.init(0); L = mz.lib()
L.mzk_debug_tail_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in ([int(a) for a in sys.argv[1:]] or [10, 20]):
    n = 1 << lg
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device_ex(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), 1, ctypes.byref(h), st) == 0
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    for _ in range(3):
        assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st) == 0
        torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert L.mzk_debug_tail_trace(buf) == 0
    t0 = buf[0]
    print("== commit 2^%d: k_reduce_tail_row phases, us after kernel entry" % lg)
    for i in sorted(range(1, 61), key=lambda i: buf[i]):
        if buf[i] >= t0 and buf[i] - t0 < 10**8:
            name = ("step t=%d" % (i - 1) if i < 39 else {39: "live entries in LDS", 40: "four Horner chains", 41: "tree level 1", 42: "tree level 2", 60: "affine stored"}.get(i, "stamp %d" % i))
            print("  %-16s %8.2f" % (name, (buf[i] - t0) / 100.0))
    if buf[61] > buf[62] > 0 and buf[60] > t0:
        print("  shader clock over the kernel: %.0f MHz" % ((buf[61] - buf[62]) / ((buf[60] - t0) / 100.0)))
    L.mzk_srs_free(h)
