"""Openings of a 2^16-leaf M128 tree: 192 single mzk_merkle_open calls vs one mzk_merkle_open_batch (FRI query phase: 64 tests x 3).
python tools/timing/merkle_open_batch.py"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, orc
import myzkp_amd as mz
mz.init(0)
for lg in (12, 16, 20):
    n = 1 << lg
    t = mz.MerkleTree(orc.M128, orc.synth_vector(orc.M128, 5, n))
    rnd = random.Random(1)
    idx = [rnd.randrange(n) for _ in range(192)]
    t.open(0); t.open_many(idx[:4])
    t0 = time.perf_counter(); a = [t.open(i) for i in idx]; t1 = time.perf_counter()
    b = t.open_many(idx); t2 = time.perf_counter()
    print("2^%d leaves, 192 openings: one by one %.2f ms, one batch %.2f ms, identical: %s" % (lg, (t1 - t0) * 1e3, (t2 - t1) * 1e3, a == b), flush=True)
    t.close()
