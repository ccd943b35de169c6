#!/bin/bash
# round 3, job f: latency of one row-cooperative doubling vs one plain single-lane doubling (kernel durations of the self-test's
# run / check kernels with 2000 dependent doublings)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03f_lat -- python3 $R/tools/timing/row_op_latency.py > $O/r03f_lat.log 2>&1
python3 - <<'PY' > $O/r03f_row_op_latency.txt
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
f = glob.glob(O + "/r03f_lat/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"]
    if "rowtest_run" in nm or "rowtest_check" in nm:
        acc[(nm.split("(")[0].split("::")[-1], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    v = sorted(acc[k])
    print("%-18s grid %8s  launches %3d  min %9.1f us  median %9.1f us  max %9.1f us" % (k[0], k[1], len(v), v[0], v[len(v)//2], v[-1]))
PY
cat $O/r03f_lat.log | grep -v amdgpu.ids; cat $O/r03f_row_op_latency.txt
