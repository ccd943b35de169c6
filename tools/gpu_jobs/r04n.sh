#!/bin/bash
# round 3, job 4n: row-cooperative product with three independent multiply-add chains per column: row EC self-tests, latency A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_row_ec.py tests/test_gpu_msm.py -x -q ) > $O/r04n_pytest.log 2>&1
grep -E "passed|failed|error" $O/r04n_pytest.log | tail -3
for rep in 1 2; do
  echo "== previous library (run $rep)" | tee -a $O/r04n_row_ilp_ab.txt
  MZK_HIP_LIB=$R/scratch_whatif/prev/libmzk_hip.so python tools/timing/small_latency.py 10,14 2>&1 | grep -v amdgpu.ids | tee -a $O/r04n_row_ilp_ab.txt
  MZK_HIP_LIB=$R/scratch_whatif/prev/libmzk_hip.so python tools/timing/generic_phases.py 20 2>&1 | grep -v amdgpu.ids | tee -a $O/r04n_row_ilp_ab.txt
  MZK_HIP_LIB=$R/scratch_whatif/prev/libmzk_hip.so python tools/timing/fold_one.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r04n_row_ilp_ab.txt
  echo "== this tree (run $rep)" | tee -a $O/r04n_row_ilp_ab.txt
  python tools/timing/small_latency.py 10,14 2>&1 | grep -v amdgpu.ids | tee -a $O/r04n_row_ilp_ab.txt
  python tools/timing/generic_phases.py 20 2>&1 | grep -v amdgpu.ids | tee -a $O/r04n_row_ilp_ab.txt
  python tools/timing/fold_one.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r04n_row_ilp_ab.txt
done
