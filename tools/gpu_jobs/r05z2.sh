#!/bin/bash
# round 5, job z2: uniform 2^20 / 2^24 commits and the generic MSM before and after the shared-bucket heavy combine, same box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05z2}
mkdir -p $O
cd $R
rm -f $O/${T}_ab.txt
for rep in 1 2 3; do
for lib in libmzk_hip_prev.so libmzk_hip.so; do
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  for lg in 20 24; do MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/commit_only.py $lg 40 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt; done
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/skew_msm.py 2>&1 | grep -v amdgpu | tail -8 | tr '\n' ';' >> $O/${T}_ab.txt; echo >> $O/${T}_ab.txt
done
done
cat $O/${T}_ab.txt
