"""mzk_fast_zerofier over the first 2^lg - 4 points of the subgroup of order 2^lg (FastStark's transition zerofier, fast_stark.rs:53-57), host buffers in and out:
    python tools/timing/zerofier_prefix_time.py [lg = 14]        (tuning build + MZK_INTERP_PREFIX=0: the subproduct tree on the same domain)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc, myzkp_amd as mz
mz.init(0)
fid = orc.M128
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = (1 << lg) - 4
om = orc.root_of(fid, lg)
e1 = np.zeros((1 << lg, 2), dtype=np.uint64); e1[1, 0] = 1
dom = np.ascontiguousarray(mz.ntt(fid, om, e1)[:n])
z = mz.fast_zerofier(fid, dom, om, 1 << lg)
best = 1e9
for _ in range(8):
    t0 = time.perf_counter(); z = mz.fast_zerofier(fid, dom, om, 1 << lg); best = min(best, (time.perf_counter() - t0) * 1e3)
zv = mz.ntt(fid, om, np.ascontiguousarray(z))
print("fast_zerofier of 2^%d - 4 subgroup points: %.3f ms per call; zero on exactly those points: %s" % (lg, best, bool(not zv[:n].any() and all(zv[i].any() for i in range(n, 1 << lg)))))
