#!/bin/bash
# round 5, job ai: the largest sizes the ABI states, on the final tree (sort slices, segment length and write-through rule all changed this round)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ai}
mkdir -p $O
cd $R
timeout 1500 python tools/timing/max_sizes.py msm 26 27 ntt 26 28 ntt_m128 28 2>&1 | grep -v amdgpu > $O/${T}_max_sizes.txt
cat $O/${T}_max_sizes.txt
