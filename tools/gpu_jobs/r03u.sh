#!/bin/bash
# round 3, job u: small commits without the sort launch, digits by the biased-window extraction (no serial carry walk, static
# word indices): suite, small sizes with and without
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r03u_pytest.log 2>&1
grep -E "passed|failed|error" $O/r03u_pytest.log | tail -3
echo "== shipped (MZK_SMALL_SCAN=1)" | tee $O/r03u_small_latency.txt
python tools/timing/small_latency.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r03u_small_latency.txt
echo "== MZK_SMALL_SCAN=0 (k_small_sort + k_small_accumulate)" | tee -a $O/r03u_small_latency.txt
MZK_SMALL_SCAN=0 python tools/timing/small_latency.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r03u_small_latency.txt
