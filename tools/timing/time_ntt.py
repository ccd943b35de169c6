import ctypes, sys, time, os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np, torch
import myzkp_amd as mz
import orc
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for fid, name, nl in ((0, "Fr", 4), (1, "M128", 2)):
    for lg in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (19, 20, 21, 22, 24)):
        n = 1 << lg
        vin = torch.empty(n*nl, dtype=torch.int64, device=dev); vout = torch.empty(n*nl, dtype=torch.int64, device=dev)
        L.mzk_synth_field_dev(fid, ctypes.c_uint64(5), ctypes.c_size_t(n), ctypes.c_void_p(vin.data_ptr()), st)
        root = mz.to_limbs([mz.root_of_unity(fid, lg)], nl)
        def run(inv=0, src=vin, dst=vout):
            rc = L.mzk_ntt_dev(fid, root.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_size_t(n), inv, st)
            assert rc == 0, L.mzk_last_error()
        run(); run(); torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        reps = 200 if lg <= 16 else 40 if lg <= 20 else 8
        ev0.record()
        for _ in range(reps): run()
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        # check vs oracle at 2^20 only, roundtrip at all
        ok = ""
        if lg in (20, 21):
            v = orc.synth_vector(fid, 5, n, 64)
            rc, want = orc.ntt_fast(fid, mz.from_limbs(root)[0], v, threads=64)
            ok = "exact=%s" % bool(np.array_equal(want.view(np.int64).reshape(-1), vout.cpu().numpy()))
        back = torch.empty_like(vin); run(1, vout, back); torch.cuda.synchronize()
        S = 32 if fid == 0 else 16
        print("%s 2^%d: %.4f ms  %.3g elems/s  hbm_frac=%.3f  roundtrip=%s %s" % (name, lg, ms, n/ms*1e3, 2*S*n/ms/1e6/8000, bool(torch.equal(back, vin)), ok), flush=True)
