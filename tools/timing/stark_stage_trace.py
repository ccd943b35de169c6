"""One stage of the STARK commit pipeline, repeated, for rocprofv3 --kernel-trace: python tools/timing/stark_stage_trace.py <interp|fri|merkle> [log2 trace = 14] [registers = 16] [reps = 4]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
import myzkp_amd as mz
mz.init(0)
what = sys.argv[1]
lgt = int(sys.argv[2]) if len(sys.argv) > 2 else 14
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
fid, p = orc.M128, orc.MOD[orc.M128]
cycles = (1 << lgt) - 3
lg_fri = lgt + 2
omicron, omega = orc.root_of(fid, lgt), orc.root_of(fid, lg_fri)
dom, acc = [], 1
for _ in range(cycles):
    dom.append(acc); acc = acc * omicron % p
dom = orc.to_limbs(dom, 2)
trace = np.stack([orc.synth_vector(fid, 100 + r, cycles) for r in range(R)])
cw = orc.synth_vector(fid, 9, 1 << lg_fri)
cws = np.stack([orc.synth_vector(fid, 200 + r, 1 << lg_fri) for r in range(R)])
def challenge(rnd, last, root):
    return None if last else int.from_bytes(hashlib.sha3_256(root + bytes([rnd])).digest(), "little") % p
for rep in range(reps):
    t0 = time.perf_counter()
    if what == "interp":
        mz.fast_interpolate_batch(fid, dom, trace, omicron, 1 << lgt)
    elif what == "fri":
        out = mz.fri_commit(fid, cw, omega, orc.M128_GEN, lg_fri - 4, challenge, keep_trees=True)
        for t in out[2]:
            if t is not None: t.close()
    else:
        mz.merkle_commit_field_batch(fid, cws)
    print("%s rep %d: %.3f ms" % (what, rep, (time.perf_counter() - t0) * 1e3), flush=True)
