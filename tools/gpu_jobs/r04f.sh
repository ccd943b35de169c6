#!/bin/bash
# round 3, job 4f: coarse sort passes with the compile-time digit walker (merged layout, 16 / 17-bit windows): MSM tests, then A/B
# against the previous library on one box (phases from the library's profiler)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_e2e_kzg.py tests/test_gpu_fuzz_slice.py -x -q ) > $O/r04f_pytest.log 2>&1
grep -E "passed|failed|error" $O/r04f_pytest.log | tail -3
for rep in 1 2; do
  echo "== previous library (run $rep)" | tee -a $O/r04f_digit_walker_ab.txt
  MZK_HIP_LIB=$R/scratch_whatif/prev/libmzk_hip.so python tools/timing/window_sweep.py 16,18,20,22 1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04f_digit_walker_ab.txt
  echo "== this tree (run $rep)" | tee -a $O/r04f_digit_walker_ab.txt
  python tools/timing/window_sweep.py 16,18,20,22 1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04f_digit_walker_ab.txt
done
