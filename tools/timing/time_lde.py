"""fast_coset_evaluate (ntt.rs:254-269) timings, device-resident: single extensions and batches (blow-up 4)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for fid, name, nl in ((0, "Fr", 4), (1, "M128", 2)):
    for lgc, batch in ((10, 1), (14, 1), (14, 64), (18, 1), (18, 8), (20, 1)):
        lgo = lgc + 2
        nc, order = 1 << lgc, 1 << lgo
        c = torch.empty(batch * nc * nl, dtype=torch.int64, device=dev)
        out = torch.empty(batch * order * nl, dtype=torch.int64, device=dev)
        L.mzk_synth_field_dev(fid, ctypes.c_uint64(3), ctypes.c_size_t(batch * nc), ctypes.c_void_p(c.data_ptr()), st)
        g = mz.to_limbs([mz.root_of_unity(fid, lgo)], nl); off = mz.to_limbs([3], nl)
        def run():
            rc = L.mzk_coset_lde_batch_dev(fid, ctypes.c_void_p(c.data_ptr()), ctypes.c_size_t(nc), off.ctypes.data_as(ctypes.c_void_p), g.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_void_p(out.data_ptr()), ctypes.c_size_t(order), ctypes.c_size_t(batch), st)
            assert rc == 0, L.mzk_last_error()
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 100 if lgo <= 18 else 20
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%s LDE 2^%d -> 2^%d x %d: %.4f ms per call, %.3g out elems/s" % (name, lgc, lgo, batch, ms, batch * order / ms * 1e3), flush=True)
