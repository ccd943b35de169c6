#!/bin/bash
# round 3, job 4r: quad addition skips its exceptional-case select chains when no quad of the wave needs one: tests + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_row_ec.py tests/test_gpu_e2e_kzg.py tests/test_gpu_fuzz_slice.py -x -q ) > $O/r04r_pytest.log 2>&1
grep -E "passed|failed|error" $O/r04r_pytest.log | tail -3
for rep in 1 2 3; do
  echo "== previous library (run $rep)" | tee -a $O/r04r_quad_selects_ab.txt
  MZK_HIP_LIB=$R/scratch_whatif/prev/libmzk_hip.so python tools/timing/window_sweep.py 16,20 1 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $O/r04r_quad_selects_ab.txt
  MZK_HIP_LIB=$R/scratch_whatif/prev/libmzk_hip.so python tools/timing/small_latency.py 10,12 2>&1 | grep -v amdgpu.ids | tee -a $O/r04r_quad_selects_ab.txt
  echo "== this tree (run $rep)" | tee -a $O/r04r_quad_selects_ab.txt
  python tools/timing/window_sweep.py 16,20 1 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $O/r04r_quad_selects_ab.txt
  python tools/timing/small_latency.py 10,12 2>&1 | grep -v amdgpu.ids | tee -a $O/r04r_quad_selects_ab.txt
done
