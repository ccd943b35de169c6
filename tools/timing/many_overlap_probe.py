"""What would splitting one many-commit batch over G streams buy?  G contexts on one GPU, each takes count / G polynomials of
the batch on its own stream (mzk_kzg_commit_srs_many_dev per context), all enqueued back to back, one synchronise at the end.
    python tools/timing/many_overlap_probe.py [lg_n:count[:width],...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
L = mz.lib()
L.mzk_ctx_stream.restype = ctypes.c_void_p
dev = torch.device("cuda", 0)
shapes = sys.argv[1] if len(sys.argv) > 1 else "10:256,10:256:10,12:64,14:16"
for spec in shapes.split(','):
    f = [int(x) for x in spec.split(':')]
    lg, count, width = f[0], f[1], (f[2] if len(f) > 2 else 1)
    n = 1 << lg
    line = []
    for G in (1, 2, 4, 8):
        mz.init_devices([0] * G)
        st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        sc = torch.empty(n * count * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
        L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n * count), ctypes.c_void_p(sc.data_ptr()), st0)
        L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st0)
        h = ctypes.c_void_p()
        assert L.mzk_srs_from_device_ex(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), width, ctypes.byref(h), st0) == 0, L.mzk_last_error()
        out = torch.zeros(count * 8, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        per = count // G
        def run():
            for g in range(G):
                assert L.mzk_ctx_select(g) == 0
                sg = ctypes.c_void_p(L.mzk_ctx_stream(g))
                assert L.mzk_kzg_commit_srs_many_dev(h, ctypes.c_void_p(sc.data_ptr() + g * per * n * 32), ctypes.c_size_t(n), ctypes.c_size_t(per),
                                                     ctypes.c_void_p(out.data_ptr() + g * per * 64), sg) == 0, L.mzk_last_error()
            L.mzk_ctx_select(0)
        for _ in range(3): run()
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize()
        line.append("G=%d %.3f ms" % (G, (time.perf_counter() - t0) / reps * 1e3))
        L.mzk_srs_free(h)
    print("%4d x 2^%-2d (%s): %s" % (count, lg, "default" if width == 1 else "%d-bit" % width, "   ".join(line)), flush=True)
mz.init_devices([0])
