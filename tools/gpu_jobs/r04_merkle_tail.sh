cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/timing/time_merkle.py 2>&1 | grep -v amdgpu
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/tools/timing/time_merkle.py > /dev/null 2>&1
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "merkle" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
