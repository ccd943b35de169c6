import sys, time, hashlib; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, orc, myzkp_amd as mz
mz.init(0)
p = orc.MOD[orc.M128]
for lg, rounds in ((12, 8), (16, 12), (20, 16)):
    cw = orc.synth_vector(orc.M128, 5, 1 << lg)
    om = orc.root_of(orc.M128, lg)
    ch = lambda r, last, root: None if last else int.from_bytes(hashlib.sha3_256(root).digest(), "little") % p
    mz.fri_commit(orc.M128, cw, om, orc.M128_GEN, rounds, ch)
    t0 = time.perf_counter()
    for _ in range(3): mz.fri_commit(orc.M128, cw, om, orc.M128_GEN, rounds, ch)
    print("fri_commit 2^%d, %d rounds: %.2f ms per call (host buffers in/out)" % (lg, rounds, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
