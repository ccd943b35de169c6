#!/bin/bash
# round 3, job 4e: differential fuzz of the C ABI against the oracle on the final tree (new: sortless small commits, tail re-cut,
# four-row inversion, fused NTT edges, Shoup twiddles): three seeds, ~4 minutes each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for seed in 301 302 303; do
  timeout 400 python tools/fuzz/differential.py 230 $seed 2>&1 | grep -v amdgpu.ids | tee -a $O/r04e_differential_fuzz.txt
done
