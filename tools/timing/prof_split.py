"""Kernel time per MSM from a rocprofv3 kernel trace: python tools/timing/prof_split.py <..._kernel_trace.csv>
An MSM ends at its k_reduce_tail[_row] when that runs ONE workgroup (merged layout: the tail writes the result itself), or
at the k_window_combine that follows a multi-workgroup k_reduce_tail (generic layout: one bucket set per window).
Kernels outside any MSM (NTT, Merkle, table building, synthetic inputs) are skipped."""
import csv, sys, collections
SKIP = ("k_ntt", "merkle", "coset", "k_gen_", "k_synth", "k_srs_", "k_xyzz_batch", "k_fb_", "k_alpha", "k_pointwise", "k_fri", "vectorized_elementwise", "k_suffix", "k_horner")
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
out = collections.defaultdict(lambda: collections.defaultdict(list))
cur, pending_generic = [], False
for r in rows:
    nm = r["Kernel_Name"].replace("mzk::", "").replace("void ", "").split("(")[0]
    if any(s in nm for s in SKIP) or nm.startswith("__amd_rocclr_copy"):
        continue
    cur.append((nm, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    kind = None
    if nm in ("k_reduce_tail", "k_reduce_tail_row"):
        if int(r["Grid_Size_X"]) <= (1024 if nm.endswith("_row") else 512):      # one workgroup
            kind = "merged (KZG commit against SRS tables)"
        else:
            pending_generic = True
    elif nm in ("k_window_combine", "k_window_combine_row") and pending_generic:
        kind, pending_generic = "generic (arbitrary points, GLV)", False
    if kind:
        agg = collections.defaultdict(float)
        for n, d in cur:
            agg[n] += d
        for n, d in agg.items():
            out[kind][n].append(d)
        cur = []
for kind in sorted(out):
    nmsm = max(len(v) for v in out[kind].values())
    print("==", kind, "--", nmsm, "MSMs (all sizes launched by the command)")
    tot = 0
    for n, v in sorted(out[kind].items(), key=lambda kv: -sum(kv[1]) / nmsm):
        print("  %-34s %5d  %8.1f us per MSM" % (n, len(v), sum(v) / nmsm))
        tot += sum(v) / nmsm
    print("  total kernel time %.1f us" % tot)
