"""Wall time of fast_zerofier / fast_evaluate / fast_interpolate through the host-buffer ABI (PCIe copies included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import orc, myzkp_amd as mz
mz.init(0)
for fid, name in ((orc.M128, "M128"), (orc.FR, "Fr")):
    for lg in (10, 14, 16, 18, 20):
        n = 1 << lg
        root, order = orc.root_of(fid, lg + 1), 1 << (lg + 1)
        dom = orc.synth_vector(fid, 1, n); vals = orc.synth_vector(fid, 2, n)
        res = []
        for fn in (lambda: mz.fast_zerofier(fid, dom, root, order), lambda: mz.fast_evaluate(fid, vals, dom, root, order),
                   lambda: mz.fast_interpolate(fid, dom, vals, root, order)):
            fn()
            t0 = time.perf_counter(); fn(); res.append((time.perf_counter() - t0) * 1e3)
        print("%-4s 2^%-2d points: zerofier %8.2f ms   evaluate %8.2f ms   interpolate %8.2f ms" % (name, lg, *res), flush=True)
