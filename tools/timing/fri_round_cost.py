import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import myzkp_amd as mz, orc
mz.init(0)
fid = orc.M128; p = orc.MOD[fid]
def challenge(rnd, last, root):
    return None if last else (int.from_bytes(root[:15], "little") + rnd) % p
for lg in (10, 14, 18):
    n = 1 << lg; rounds = lg - 5
    cw = orc.synth_vector(fid, lg, n); omega = orc.root_of(fid, lg)
    for keep in (False, True):
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            out = mz.fri_commit(fid, cw, omega, orc.M128_GEN, rounds, challenge, keep_trees=keep)
            dt = (time.perf_counter() - t0) * 1e3
            if keep:
                for t in out[2]:
                    if t is not None: t.close()
            best = min(best, dt)
        print("2^%d, %d rounds, keep_trees=%s: %.3f ms (%.0f us per round)" % (lg, rounds, keep, best, best / rounds * 1e3), flush=True)
