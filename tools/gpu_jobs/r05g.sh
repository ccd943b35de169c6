#!/bin/bash
# round 5, job g: new memory tests, the bench line with the reworked sub-leg timing + STARK pipeline + copy kernel, FRI round costs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05g}
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_dev_api.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
timeout 1200 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "bench rc=$?" >> $O/${T}_pytest.log
python tools/timing/fri_round_cost.py > $O/${T}_fri_round_cost.txt 2>&1
python tools/timing/stark_commit_pipeline.py 12 16 > $O/${T}_stark_pipeline_2pow12.txt 2>&1
tail -5 $O/${T}_pytest.log; tail -3 $O/${T}_bench.err; python3 - <<PY
import json
d = json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
for k in ("ntt", "ntt_m128", "coset_lde_m128", "merkle_m128", "msm_generic"):
    print(k, d[k]["ms_per_step"], d[k].get("ms_per_step_with_event_pair"), d[k].get("ms_per_step_without_event_pair"))
print(json.dumps(d["stark_commit_pipeline"], indent=1)[:1500])
print(d["hbm_copy"]); print(d["alu_roofline"])
print(d["extra_sizes_1gpu"])
PY
grep -v amdgpu $O/${T}_fri_round_cost.txt; grep -v amdgpu $O/${T}_stark_pipeline_2pow12.txt
