#!/bin/bash
# round 3, job n: A/B of 512 coarse sort bins (one more bit for the point reference: 4-byte records at 2^20 with 17-bit windows)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do
  echo "== shipped (256 coarse bins), run $rep" >> $O/r03n.txt
  timeout 600 python tools/timing/window_sweep.py 16,18,20,22 1 >> $O/r03n.txt 2>&1
  timeout 600 python tools/timing/small_latency.py 14,18,20 >> $O/r03n.txt 2>&1
  echo "== 512 coarse bins, run $rep" >> $O/r03n.txt
  MZK_HIP_LIB=$R/scratch_whatif/cb9/libmzk_hip.so timeout 600 python tools/timing/window_sweep.py 16,18,20,22 1 >> $O/r03n.txt 2>&1
  MZK_HIP_LIB=$R/scratch_whatif/cb9/libmzk_hip.so timeout 600 python tools/timing/small_latency.py 14,18,20 >> $O/r03n.txt 2>&1
done
MZK_HIP_LIB=$R/scratch_whatif/cb9/libmzk_hip.so timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_dev_api.py -m gpu -x -q 2>&1 | tail -3 >> $O/r03n.txt
grep -v amdgpu.ids $O/r03n.txt
