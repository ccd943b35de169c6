#!/bin/bash
# round 3, job 4b: default bench line after the tail / sortless / NTT work
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python bench.py > $O/r04b_bench.json 2> $O/r04b_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04b_bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","cpu_baseline"): print(k, d.get(k))
for k,v in d.items():
    if k not in ("value","ms_per_step","roofline","cpu_baseline","config") : print(k, json.dumps(v)[:300])
PY
