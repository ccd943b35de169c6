#!/bin/bash
# round 6, job b: several halving steps per launch (k_halve_multi) and 4-byte sort records through 512 coarse bins at 17-bit windows:
# parity first, then same-box A/B of the four combinations through the tuning build's switches, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_many.py tests/test_gpu_full_size.py tests/test_gpu_dev_api.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r06b_pytest.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
rm -f $O/r06b_ab.txt
for rep in 1 2 3; do
  for combo in "0 8" "1 8" "0 9" "1 9"; do
    set -- $combo
    echo "== MZK_HALVE_MULTI=$1 MZK_COARSE_LOG_17=$2 (rep $rep)" >> $O/r06b_ab.txt
    MZK_HALVE_MULTI=$1 MZK_COARSE_LOG_17=$2 python tools/timing/commit_only.py 20 40 2>&1 | grep -v amdgpu.ids >> $O/r06b_ab.txt
  done
done
for combo in "0 8" "1 9"; do
  set -- $combo
  echo "== phases MZK_HALVE_MULTI=$1 MZK_COARSE_LOG_17=$2" >> $O/r06b_ab.txt
  MZK_HALVE_MULTI=$1 MZK_COARSE_LOG_17=$2 python tools/timing/window_sweep.py 16,18,20,22,24 1 2>&1 | grep -v amdgpu.ids | cut -c1-220 >> $O/r06b_ab.txt
  MZK_HALVE_MULTI=$1 python tools/timing/generic_phases.py 16 20 24 2>&1 | grep -v amdgpu.ids >> $O/r06b_ab.txt
  MZK_HALVE_MULTI=$1 python tools/timing/small_latency.py 10,12,14 2>&1 | grep -v amdgpu.ids >> $O/r06b_ab.txt
  MZK_HALVE_MULTI=$1 python tools/timing/many_commit.py 2>&1 | grep -v amdgpu.ids | tail -12 >> $O/r06b_ab.txt
done
cat $O/r06b_ab.txt
