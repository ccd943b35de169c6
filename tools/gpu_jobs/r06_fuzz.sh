#!/bin/bash
# round 6: differential fuzz of the C ABI against the oracle on the final tree (new this round: sort without global scans, 4-byte records
# through 512 bins, several halving steps per launch, 19-bit generic windows, host buffers in pieces, sharded open quotient)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/r06_differential_fuzz.txt
for seed in ${@:-6101 6102 6103 6104}; do
  timeout 400 python tools/fuzz/differential.py 230 $seed 2>&1 | grep -v amdgpu.ids | tee -a $O/r06_differential_fuzz.txt
done
