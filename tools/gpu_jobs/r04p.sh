#!/bin/bash
# round 3, job 4p: differential fuzz again on the last tree (staged coarse scatter, two-launch scans, fine-sort retune, row product)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for seed in 401 402; do
  timeout 400 python tools/fuzz/differential.py 230 $seed 2>&1 | grep -v amdgpu.ids | tee -a $O/r04p_differential_fuzz.txt
done
