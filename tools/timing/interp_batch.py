"""fast_interpolate of 16 registers over one domain: 16 single calls vs one mzk_fast_interpolate_batch call (host buffers).
python tools/timing/interp_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, orc
import myzkp_amd as mz
mz.init(0)
fid = orc.M128
for lg in (10, 12, 14, 16):
    n = (1 << lg) - 5
    om = orc.root_of(fid, lg + 1)
    dom = orc.synth_vector(fid, 3, n)
    vals = np.stack([orc.synth_vector(fid, 10 + k, n) for k in range(16)])
    mz.fast_interpolate(fid, dom, vals[0], om, 2 << lg)
    t0 = time.perf_counter()
    singles = [mz.fast_interpolate(fid, dom, vals[k], om, 2 << lg) for k in range(16)]
    t1 = time.perf_counter()
    got = mz.fast_interpolate_batch(fid, dom, vals, om, 2 << lg)
    t2 = time.perf_counter()
    same = all(np.array_equal(a, b) for a, b in zip(singles, got))
    print("M128 %6d points, 16 registers: 16 single calls %.1f ms, one batch call %.1f ms, identical: %s" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, same), flush=True)
