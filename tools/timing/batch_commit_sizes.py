"""mzk_kzg_commit_srs_batch_dev across sizes: ms per commit one at a time vs batches of 32 with 2 / 4 commits in flight.
python tools/timing/batch_commit_sizes.py [contexts, default 4]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init_devices([0] * (int(sys.argv[1]) if len(sys.argv) > 1 else 4)); L = mz.lib()
dev = torch.device("cuda", 0)
d0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 32
for lg in (10, 12, 14, 16, 18, 20):
    n = 1 << lg
    pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), d0)
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), d0) == 0
    coefs = torch.empty(B * n * 4, dtype=torch.int64, device=dev)
    for k in range(B):
        L.mzk_synth_field_dev(0, ctypes.c_uint64(1 + k), ctypes.c_size_t(n), ctypes.c_void_p(coefs.data_ptr() + k * n * 32), d0)
    out = torch.zeros(B * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    row = []
    for lanes in ((1, 2, 4, 8) if mz.ctx_count() >= 8 else (1, 2, 4)):
        def one():
            assert L.mzk_kzg_commit_srs_batch_dev(h, ctypes.c_void_p(coefs.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(B), ctypes.c_void_p(out.data_ptr()), ctypes.c_int(lanes), d0) == 0
        one(); one(); torch.cuda.synchronize()
        reps = 6 if lg >= 18 else 20
        t0 = time.perf_counter()
        for _ in range(reps): one()
        torch.cuda.synchronize()
        row.append("%d in flight %.3f ms" % (lanes, (time.perf_counter() - t0) / (reps * B) * 1e3))
    print("2^%-2d coefficients, per commit: " % lg + " | ".join(row), flush=True)
    L.mzk_srs_free(h)
    del pt, coefs; torch.cuda.empty_cache()
