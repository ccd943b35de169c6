"""A/B of k_seg_accumulate variants on one box: PREFETCH on/off x segment length, merged (SRS tables) and generic MSM at
2^LOG pairs.  Each configuration runs in its own process (the knobs are read once): python tools/timing/acc_sweep.py [LOG]
    child: python tools/timing/acc_sweep.py --child LOG   (MZK_ACC_PREFETCH / MZK_ACC_SEG in the environment)"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(lg):
    import torch
    import myzkp_amd as mz
    mz.init(0); L = mz.lib()
    dev = torch.device("cuda", 0); n = 1 << lg
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n), ctypes.c_void_p(sc.data_ptr()), st)
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.byref(h), st) == 0
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    res = {}
    for name, fn in (("merged", lambda: L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()), 0, st)),
                     ("generic", lambda: L.mzk_msm_g1_bn254_dev(ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr() + 64), st))):
        t_end = time.perf_counter() + 0.6          # settle: clocks ramp for a few hundred ms
        while time.perf_counter() < t_end:
            assert fn() == 0
            torch.cuda.synchronize()
        L.mzk_prof_reset(); L.mzk_prof_enable(1)
        reps = 20
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e3
        L.mzk_prof_enable(0)
        ms, cnt = ctypes.c_double(), ctypes.c_uint64()
        L.mzk_prof_read(2, ctypes.byref(ms), ctypes.byref(cnt))
        res[name] = (dt, ms.value / max(cnt.value, 1))
    print("RESULT prefetch=%s seg=%s merged %.3f ms (accumulate+combine %.3f) generic %.3f ms (accumulate+combine %.3f) result %s" % (
        os.environ.get("MZK_ACC_PREFETCH", "-"), os.environ.get("MZK_ACC_SEG", "auto"), res["merged"][0], res["merged"][1],
        res["generic"][0], res["generic"][1], hex(int(out[0].item()) & 0xffffffff)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
        E = 16 << lg
        segs = {0: ["0"] + [str(-(-E // (256 * 4 * 4 * 64 * k))) for k in (2,)] + ["48", "96"],
                1: ["0"] + [str(-(-E // (256 * 4 * 3 * 64 * k))) for k in (2,)] + ["64"]}
        for pf in (0, 1):
            for seg in segs[pf]:
                env = dict(os.environ, MZK_ACC_PREFETCH=str(pf))
                if seg != "0":
                    env["MZK_ACC_SEG"] = seg
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(lg)], env=env, capture_output=True, text=True, timeout=600)
                print((r.stdout.strip().splitlines() or ["(no output) " + r.stderr[-300:]])[-1], flush=True)
