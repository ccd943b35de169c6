#!/bin/bash
# round 3, job y: A/B of the fused first / last NTT stage pairs on one box (MZK_NTT_FUSE_EDGES=0/1, alternating)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do
  for f in 0 1; do
    echo "== MZK_NTT_FUSE_EDGES=$f (run $rep)" | tee -a $O/r03y_ntt_fuse_ab.txt
    MZK_NTT_FUSE_EDGES=$f python tools/timing/time_ntt.py 16,18,20,22,24 2>&1 | grep -v amdgpu.ids | tee -a $O/r03y_ntt_fuse_ab.txt
  done
done
