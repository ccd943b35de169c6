"""Latency of ONE row-cooperative doubling (mzk_row.h) on an otherwise idle GPU: mzk_selftest_row_ec with one pair and R
dependent doublings, R = 0 vs R = 4000 (the self-test also runs the plain doubling chain on one lane in its checker kernel:
both chains are in the difference, so the plain chain is timed alone with the asm self-test's neighbour ... simply report the sum
and, from a second run with 64 pairs, the same).  python tools/timing/row_op_latency.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import myzkp_amd as mz
mz.init(0); L = mz.lib()
bad = ctypes.c_uint64(0)


def run(n, reps):
    for _ in range(3):
        assert L.mzk_selftest_row_ec(ctypes.c_uint64(7), ctypes.c_size_t(n), ctypes.c_int(reps), ctypes.byref(bad)) == 0 and bad.value == 0
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        L.mzk_selftest_row_ec(ctypes.c_uint64(7), ctypes.c_size_t(n), ctypes.c_int(reps), ctypes.byref(bad))
    return (time.perf_counter() - t0) / K


for n in (1, 64, 1024, 4096):
    a, b = run(n, 0), run(n, 2000)
    print("n = %5d pairs: (row dbl + plain one-lane dbl) = %.3f us per doubling pair" % (n, (b - a) / 2000 * 1e6), flush=True)
