"""Kernel-by-kernel view of ONE small commit (2^10 and 2^13 coefficients against an SRS handle) for rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/timing/small_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg in (10, 13):
    n = 1 << lg
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    for _ in range(20):
        assert L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st) == 0
        torch.cuda.synchronize()
    L.mzk_srs_free(h)
