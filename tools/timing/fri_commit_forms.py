"""FRI::commit (fri.rs:144-209) of one M128 codeword, trees kept: host codeword in / all codewords out, against the codeword
already in HBM and nothing but the roots coming back (mzk_fri_commit_keep_trees_dev, codewords_out = NULL)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import myzkp_amd as mz, orc
mz.init(0)
fid = orc.M128
p = orc.MOD[fid]
def challenge(rnd, last, root):
    return None if last else (int.from_bytes(root[:15], "little") + rnd) % p
for lg in (14, 18, 22):
    n = 1 << lg
    rounds = lg - 5
    cw = orc.synth_vector(fid, lg, n)
    d = torch.from_numpy(cw.view(np.int64).reshape(-1).copy()).to("cuda:0")
    omega = orc.root_of(fid, lg)
    res = {}
    for name, kw in (("host in, codewords out", dict()), ("device in, roots only", dict(codewords=False, device_ptr=d.data_ptr(), n=n))):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = mz.fri_commit(fid, None if "device_ptr" in kw else cw, omega, orc.M128_GEN, rounds, challenge, keep_trees=True, **kw)
            dt = (time.perf_counter() - t0) * 1e3
            for t in out[2]:
                if t is not None: t.close()
            best = min(best, dt)
            res[name] = out[1]
        print("2^%d elements, %d rounds: %-26s %.2f ms" % (lg, rounds, name, best), flush=True)
    assert len(set(tuple(v) for v in res.values())) == 1
