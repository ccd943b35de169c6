#!/bin/bash
# round 6, job i: kernel trace of the 2^20 commit on the current tree (what each of the launches around the accumulate costs now)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06i_prof -- python3 $R/tools/timing/commit_only.py 20 40 > $O/r06i_commit.log 2>&1
python3 $R/tools/timing/prof_split.py $(find $O/r06i_prof -name "*kernel_trace.csv" | head -1) > $O/r06i_budget.txt 2>&1
python3 - <<'PY' > $O/r06i_one_commit_trace.txt
import csv, glob, os
f = glob.glob(os.environ.get("O", "/root/repo/gpurun_out") + "/r06i_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last commit of the run: from its coarse count to its tail
idx = [i for i, r in enumerate(rows) if "k_coarse_count" in r["Kernel_Name"]]
a = idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:a + 14]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s start %8.1f us  dur %7.1f us  gap %5.1f us  grid %s" % (r["Kernel_Name"].replace("mzk::", "")[:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size")))
    prev_end = e
PY
find $O/r06i_prof -name "*.csv" -size +4M -delete
cat $O/r06i_budget.txt | head -30; cat $O/r06i_one_commit_trace.txt
