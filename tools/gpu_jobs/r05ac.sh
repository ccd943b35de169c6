#!/bin/bash
# round 5, job ac: which sort kernels carry the skew (bytes / 248-bit / bits scalars), per-kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ac}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pat in uniform bytes 248-bit bits; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_$pat -- python3 $R/tools/timing/skew_msm.py $pat > $O/${T}_$pat.log 2>&1
  find $O/${T}_$pat -name "*kernel_stats.csv" -exec cp {} $O/${T}_${pat}_kernel_stats.csv \;
  echo "== $pat"; python3 - <<PY
import csv
for r in csv.DictReader(open("$O/${T}_${pat}_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("coarse", "fine", "scan")): print("  %-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $O -name "*.csv" -size +4M -delete
