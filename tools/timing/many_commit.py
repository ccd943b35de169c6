"""Grid-batched commitments of many short polynomials (mzk_kzg_commit_srs_many_dev) against the same batch one commit at a
time: ms per batch, us per commit, the per-phase split of the pass.
    python tools/timing/many_commit.py [lg_n:count[:width[:direct[:bits]]],...]      width: 1 = default window tables, 8..22 that width;
direct: 0 = none, 8..12 = direct tables of that width (mzk_srs_build_direct); bits: coefficients below 2^bits (the reference's DAS
callers commit to 31-byte chunks: 248), default full-width"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
PH = {}
for i in range(32):
    L.mzk_prof_name.restype = ctypes.c_char_p
    nm = L.mzk_prof_name(i)
    if nm: PH[i] = nm.decode()
shapes = sys.argv[1] if len(sys.argv) > 1 else "10:256,12:64,8:1024,14:16,10:256:8,10:256:11"
for spec in shapes.split(','):
    f = [int(x) for x in spec.split(':')]
    lg, count, width, direct, bits = f[0], f[1], (f[2] if len(f) > 2 else 1), (f[3] if len(f) > 3 else 0), (f[4] if len(f) > 4 else 256)
    n = 1 << lg
    sc = torch.empty(n * count * 4, dtype=torch.int64, device=dev); pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    L.mzk_synth_field_dev(0, ctypes.c_uint64(1), ctypes.c_size_t(n * count), ctypes.c_void_p(sc.data_ptr()), st)
    if bits < 256:
        assert bits > 192
        sc.view(-1, 4)[:, 3] &= (1 << (bits - 192)) - 1
    L.mzk_synth_g1_points_dev(ctypes.c_uint64(2), ctypes.c_size_t(n), ctypes.c_void_p(pt.data_ptr()), st)
    h = ctypes.c_void_p()
    assert L.mzk_srs_from_device_ex(ctypes.c_void_p(pt.data_ptr()), ctypes.c_size_t(n), width, ctypes.byref(h), st) == 0, L.mzk_last_error()
    note = ""
    if direct:
        L.mzk_srs_table_bytes.restype = ctypes.c_size_t
        b0 = L.mzk_srs_table_bytes(h)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert L.mzk_srs_build_direct(h, direct, ctypes.c_size_t(64 << 30), st) == 0, L.mzk_last_error()
        note = " + %d-bit direct tables (%.2f GiB, built in %.1f ms)" % (direct, (L.mzk_srs_table_bytes(h) - b0) / 2**30, (time.perf_counter() - t0) * 1e3)
    out = torch.zeros(count * 8, dtype=torch.int64, device=dev); out1 = torch.zeros(count * 8, dtype=torch.int64, device=dev)
    many = lambda: L.mzk_kzg_commit_srs_many_dev(h, ctypes.c_void_p(sc.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(count), ctypes.c_void_p(out.data_ptr()), st)
    def loop():
        for k in range(count):
            rc = L.mzk_kzg_commit_srs_dev(h, ctypes.c_void_p(sc.data_ptr() + k * n * 32), ctypes.c_size_t(n), ctypes.c_void_p(out1.data_ptr() + k * 64), 0, st)
            if rc: return rc
        return 0
    res = []
    for fn, reps in ((many, 30), (loop, 3)):
        for _ in range(3):
            assert fn() == 0, L.mzk_last_error()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e3)
    same = bool(torch.equal(out, out1))
    L.mzk_prof_reset(); L.mzk_prof_enable(1)
    for _ in range(10): many()
    torch.cuda.synchronize()
    split = []
    for i, nm in PH.items():
        ms = ctypes.c_double(); cnt = ctypes.c_uint64()
        L.mzk_prof_read(i, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value: split.append("%s %.3f" % (nm, ms.value / 10))
    L.mzk_prof_enable(0)
    print("%4d x 2^%-2d%s (tables: %s): many %.3f ms = %.2f us per commit | one at a time %.3f ms = %.1f us per commit | %.1fx | same points: %s | %s"
          % (count, lg, "" if bits == 256 else " %d-bit coefficients" % bits, ("default" if width == 1 else "%d-bit" % width) + note, res[0], res[0] / count * 1e3, res[1], res[1] / count * 1e3, res[1] / res[0], same, "  ".join(split)), flush=True)
    L.mzk_srs_free(h)
