#!/bin/bash
# round 3, job o: k_seg_accumulate now needs 140 VGPRs (3 waves per SIMD) but its grid is sized for 4: A/B of (a) shipped,
# (b) __launch_bounds__(256, 4) = 128 VGPRs, (c) segments sized for 3 waves per SIMD (MZK_ACC_SEG)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do
  echo "== (a) shipped, run $rep" >> $O/r03o.txt
  timeout 300 python tools/timing/window_sweep.py 20,22 1 >> $O/r03o.txt 2>&1
  echo "== (b) launch_bounds(256, 4), run $rep" >> $O/r03o.txt
  MZK_HIP_LIB=$R/scratch_whatif/lb4/libmzk_hip.so timeout 300 python tools/timing/window_sweep.py 20,22 1 >> $O/r03o.txt 2>&1
  echo "== (c) segments for 3 waves per SIMD: MZK_ACC_SEG=80 at 2^20, run $rep" >> $O/r03o.txt
  MZK_ACC_SEG=80 timeout 300 python tools/timing/window_sweep.py 20 1 >> $O/r03o.txt 2>&1
  echo "== (c') MZK_ACC_SEG=320 at 2^22, run $rep" >> $O/r03o.txt
  MZK_ACC_SEG=320 timeout 300 python tools/timing/window_sweep.py 22 1 >> $O/r03o.txt 2>&1
done
grep -v amdgpu.ids $O/r03o.txt
