#!/bin/bash
# round 6, job n: the heads of the bucket reduction weighted in the last halving launch (doublings beside region 0's recursion) instead of in the
# single-workgroup tail: parity, then same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_many.py tests/test_gpu_full_size.py tests/test_gpu_dev_api.py tests/test_gpu_fuzz_slice.py tests/test_gpu_row_ec.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r06n_pytest.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
rm -f $O/r06n_ab.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== MZK_HALVE_WEIGH=$v (rep $rep)" >> $O/r06n_ab.txt
    MZK_HALVE_WEIGH=$v python tools/timing/commit_only.py 20 60 2>&1 | grep -v amdgpu.ids >> $O/r06n_ab.txt
    MZK_HALVE_WEIGH=$v python tools/timing/window_sweep.py 16,18,20,24 1 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $O/r06n_ab.txt
    MZK_HALVE_WEIGH=$v python tools/timing/generic_phases.py 16 20 2>&1 | grep -v amdgpu.ids >> $O/r06n_ab.txt
    MZK_HALVE_WEIGH=$v python tools/timing/small_latency.py 10,12,14 2>&1 | grep -v amdgpu.ids >> $O/r06n_ab.txt
  done
done
cat $O/r06n_ab.txt
