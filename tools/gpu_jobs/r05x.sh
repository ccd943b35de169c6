#!/bin/bash
# round 5, job x: HEAVY_SLOTS 16 against 32, same box: grid-batched commits (full-width and 248-bit coefficients), the 2^20 / 2^24 commits
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05x}
mkdir -p $O
cd $R
rm -f $O/${T}_ab.txt
for rep in 1 2; do
for lib in libmzk_hip_h32.so libmzk_hip.so; do
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/many_commit.py 10:256,10:256:1:0:248,8:1024:1:0:248,12:64:1:0:248,14:16:1:0:248 2>&1 | grep -v amdgpu | cut -c1-100,180-400 >> $O/${T}_ab.txt
  for lg in 20 24; do MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/commit_only.py $lg 40 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt; done; MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_msm.py 2>&1 | grep -v amdgpu | tail -3 >> $O/${T}_ab.txt
done
done
cat $O/${T}_ab.txt
