#!/bin/bash
# round 5, job ah: differential fuzz (short scalars, sizes into the two-level sort, the device-resident interpolation), two seeds
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ah}
mkdir -p $O
cd $R
for seed in ${SEEDS:-51 52}; do timeout 900 python tools/fuzz/differential.py ${FUZZ_S:-240} $seed 2>&1 | grep -v amdgpu | tail -6 >> $O/${T}_fuzz.txt; done
cat $O/${T}_fuzz.txt
