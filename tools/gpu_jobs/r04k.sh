#!/bin/bash
# round 3, job 4k: scans in two launches (block totals summed by the consumer workgroups): suite + commit timings
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r04k_pytest.log 2>&1
grep -E "passed|failed|error" $O/r04k_pytest.log | tail -3
for rep in 1 2; do
  echo "== previous library (run $rep)" | tee -a $O/r04k_scan_two_launches_ab.txt
  MZK_HIP_LIB=$R/scratch_whatif/prev/libmzk_hip.so python tools/timing/window_sweep.py 16,20,22 1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04k_scan_two_launches_ab.txt
  echo "== this tree (run $rep)" | tee -a $O/r04k_scan_two_launches_ab.txt
  python tools/timing/window_sweep.py 16,20,22 1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04k_scan_two_launches_ab.txt
done
