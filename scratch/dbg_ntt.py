import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import myzkp_amd as mz, orc
mz.init(0)
for fid in (orc.M128, orc.FR):
  for lg in range(1, 15):
    n = 1 << lg
    v = orc.synth_vector(fid, 3, n)
    w = orc.root_of(fid, lg)
    rc, want = orc.ntt_fast(fid, w, v)
    got = mz.ntt(fid, w, v)
    bad = np.nonzero((got != want).any(axis=1))[0]
    print(fid, lg, "OK" if len(bad) == 0 else "BAD count=%d first=%s" % (len(bad), bad[:8]))
