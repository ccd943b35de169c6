"""The commit side of a STARK prover's work on the M128 field, item by item (as the reference's loops do it,
fast_stark.rs:203-243,337 and fri.rs:144-260) and with the one-call forms -- same results, host buffers in and out:
  interpolate R registers over the trace domain -> low-degree-extend them -> Merkle::commit every codeword ->
  FRI::commit one codeword -> open 3 x T indices per round.
python tools/timing/stark_commit_pipeline.py [log2 trace length = 12] [registers = 16]"""
import hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
import myzkp_amd as mz

mz.init(0)
fid, p = orc.M128, orc.MOD[orc.M128]
lgt = int(sys.argv[1]) if len(sys.argv) > 1 else 12
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cycles = (1 << lgt) - 3                        # a trace that does not fill its power-of-two domain
lg_om, lg_fri = lgt, lgt + 2                   # omicron domain, FRI domain (expansion factor 4)
omicron, omega = orc.root_of(fid, lg_om), orc.root_of(fid, lg_fri)
dom, acc = [], 1
for _ in range(cycles):
    dom.append(acc); acc = acc * omicron % p
dom = orc.to_limbs(dom, 2)
trace = np.stack([orc.synth_vector(fid, 100 + r, cycles) for r in range(R)])
T, rounds = 32, lg_fri - 4


def timed(fn):
    t0 = time.perf_counter(); out = fn(); return out, (time.perf_counter() - t0) * 1e3


def challenge(rnd, last, root):
    return None if last else int.from_bytes(hashlib.sha3_256(root + bytes([rnd])).digest(), "little") % p


def pad(polys, n):
    out = np.zeros((len(polys), n, 2), dtype=np.uint64)
    for k, c in enumerate(polys):
        out[k, :len(c)] = c
    return out


best = {}
for rep in range(3):            # first pass warms plans and workspaces; the fastest of the others counts (the runtime stalls
    # for ~35 ms once in a few hundred calls, whatever is running)
    polys_loop, t_i1 = timed(lambda: [mz.fast_interpolate(fid, dom, trace[r], omicron, 1 << lg_om) for r in range(R)])
    polys_batch, t_i2 = timed(lambda: mz.fast_interpolate_batch(fid, dom, trace, omicron, 1 << lg_om))
    import torch
    d_out = torch.zeros(R * cycles * 2, dtype=torch.int64, device="cuda")
    def interp_hbm():
        d_tr = torch.from_numpy(np.ascontiguousarray(trace).view(np.int64).reshape(-1)).cuda()
        lens = mz.fast_interpolate_batch_dev(fid, dom, d_tr.data_ptr(), R, omicron, 1 << lg_om, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return lens
    lens_hbm, t_i3 = timed(interp_hbm)
    rows_hbm = d_out.cpu().numpy().view(np.uint64).reshape(R, cycles, 2)
    assert all(lens_hbm[r] == len(polys_batch[r]) and np.array_equal(rows_hbm[r, :lens_hbm[r]], polys_batch[r]) for r in range(R))
    coefs = pad(polys_batch, 1 << lg_om)
    cw_loop, t_l1 = timed(lambda: [mz.coset_lde(fid, coefs[r], orc.M128_GEN, omega, 1 << lg_fri) for r in range(R)])
    cw_batch, t_l2 = timed(lambda: mz.coset_lde_batch(fid, coefs, orc.M128_GEN, omega, 1 << lg_fri))
    roots_loop, t_m1 = timed(lambda: [mz.merkle_commit_field(fid, cw_batch[r]) for r in range(R)])
    roots_batch, t_m2 = timed(lambda: mz.merkle_commit_field_batch(fid, cw_batch))
    (cws, froots, trees), t_f = timed(lambda: mz.fri_commit(fid, cw_batch[0], omega, orc.M128_GEN, rounds, challenge, keep_trees=True))
    rnd = random.Random(7)
    idx = [[rnd.randrange(len(cws[r])) for _ in range(3 * T)] for r in range(rounds)]
    open_loop, t_o1 = timed(lambda: [[trees[r].open(i) for i in idx[r]] for r in range(rounds) if trees[r] is not None])
    open_batch, t_o3 = timed(lambda: [trees[r].open_many(idx[r]) for r in range(rounds) if trees[r] is not None])      # one call per round
    live = [r for r in range(rounds) if trees[r] is not None]
    open_multi, t_o2 = timed(lambda: mz.merkle_open_multi([trees[r] for r in live], [idx[r] for r in live]))            # one call for all rounds
    assert open_multi == open_batch
    for t in trees:
        if t is not None: t.close()
    same = (all(np.array_equal(a, b) for a, b in zip(polys_loop, polys_batch)) and all(np.array_equal(a, b) for a, b in zip(cw_loop, cw_batch))
            and roots_loop == roots_batch and open_loop == open_batch)
    assert same
    if rep == 0:
        continue
    for k, v in (("i1", t_i1), ("i2", t_i2), ("i3", t_i3), ("l1", t_l1), ("l2", t_l2), ("m1", t_m1), ("m2", t_m2), ("o1", t_o1), ("o2", t_o2), ("o3", t_o3), ("f", t_f)):
        best[k] = min(best.get(k, v), v)
print("M128, %d registers x %d cycles, FRI domain 2^%d, %d FRI rounds, %d openings per round" % (R, cycles, lg_fri, rounds, 3 * T))
print("  step                         item by item     one call")
for name, a, b in (("fast_interpolate", "i1", "i2"), ("fast_coset_evaluate", "l1", "l2"), ("Merkle::commit", "m1", "m2"), ("Merkle::open (query phase)", "o1", "o2")):
    print("  %-28s %9.2f ms  %9.2f ms" % (name, best[a], best[b]))
print("  %-28s %9s     %9.2f ms   (one call per round: mzk_merkle_open_batch)" % ("  same, round by round", "", best["o3"]))
print("  %-28s %9s     %9.2f ms   (mzk_fast_interpolate_batch_dev: trace uploaded inside, coefficients left in HBM)" % ("fast_interpolate, HBM form", "", best["i3"]))
print("  %-28s %9s     %9.2f ms   (one call in both)" % ("FRI::commit, trees kept", "", best["f"]))
print("  total                        %9.2f ms  %9.2f ms   identical results: True" % (best["i1"] + best["l1"] + best["m1"] + best["o1"] + best["f"],
                                                                                  best["i2"] + best["l2"] + best["m2"] + best["o2"] + best["f"]))
