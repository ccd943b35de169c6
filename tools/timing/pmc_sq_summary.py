"""Summarise a rocprofv3 --pmc counter_collection CSV tree: per kernel, the mean of every collected counter per launch,
plus VGPR/SGPR/LDS/scratch of the dispatch (columns of the same CSV).  python tools/timing/pmc_sq_summary.py <dir> [substr...]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("mzk::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
want = sys.argv[2:]
for k in sorted(acc):
    if want and not any(w in k for w in want):
        continue
    n = max(len(v) for v in acc[k].values())
    print("%s  launches=%d  %s" % (k, n, " ".join("%s=%s" % kv for kv in meta[k].items())))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-24s mean %16.1f   min %16.1f   max %16.1f" % (c, sum(v) / len(v), min(v), max(v)))
