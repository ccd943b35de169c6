#!/bin/bash
# round 6, job e: the whole GPU suite on the current tree (scan-free sort with two-barrier block scans, 19-bit generic windows from
# 3 x 2^21 pairs, the advisor's five items), then phases against the r05 library on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/r06e_pytest.txt
rm -f $O/r06e_ab.txt
for rep in 1 2 3; do
for lib in ab/libmzk_hip_r05.so myzkp_amd/libmzk_hip.so; do
  echo "== $lib (rep $rep)" >> $O/r06e_ab.txt
  MZK_HIP_LIB=$R/$lib python tools/timing/commit_only.py 20 60 2>&1 | grep -v amdgpu.ids >> $O/r06e_ab.txt
  MZK_HIP_LIB=$R/$lib python tools/timing/window_sweep.py 16,20,24 1 2>&1 | grep -v amdgpu.ids | cut -c1-220 >> $O/r06e_ab.txt
  MZK_HIP_LIB=$R/$lib python tools/timing/generic_phases.py 20 24 2>&1 | grep -v amdgpu.ids >> $O/r06e_ab.txt
done
done
cat $O/r06e_ab.txt
