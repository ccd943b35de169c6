#!/bin/bash
# round 3, job t: (1) wave inversion with the four numbers in the four rows (one re-cut per batch), (2) small commits without the
# sort launch (every bucket's workgroup walks the scalars): suite, fold latency, small sizes with and without (2)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r03t_pytest.log 2>&1
grep -E "passed|failed|error" $O/r03t_pytest.log | tail -3
python tools/timing/fold_one.py 2>&1 | grep -v amdgpu.ids | tee $O/r03t_fold.txt
echo "== shipped (MZK_SMALL_SCAN=1)" | tee $O/r03t_small_latency.txt
python tools/timing/small_latency.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r03t_small_latency.txt
echo "== MZK_SMALL_SCAN=0 (k_small_sort + k_small_accumulate)" | tee -a $O/r03t_small_latency.txt
MZK_SMALL_SCAN=0 python tools/timing/small_latency.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r03t_small_latency.txt
