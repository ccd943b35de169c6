#!/bin/bash
# round 3, job 4h: coarse scatter staged through LDS (coalesced runs): MSM tests, then A/B (MZK_COARSE_STAGED=0/1) on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_e2e_kzg.py tests/test_gpu_fuzz_slice.py tests/test_gpu_max_sizes.py -x -q ) > $O/r04h_pytest.log 2>&1
grep -E "passed|failed|error" $O/r04h_pytest.log | tail -3
for rep in 1 2; do for f in 0 1; do
  echo "== MZK_COARSE_STAGED=$f (run $rep)" | tee -a $O/r04h_coarse_staged_ab.txt
  MZK_COARSE_STAGED=$f python tools/timing/window_sweep.py 18,20,22,24 1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04h_coarse_staged_ab.txt
done; done
