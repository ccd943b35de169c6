#!/bin/bash
# round 5, job aa: phase split of the 2^20 MSM on short / skewed scalars
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05aa}
mkdir -p $O
cd $R
python tools/timing/skew_msm.py "uniform,bits,bytes,16-bit scalars,32-bit,64-bit,128-bit,248-bit,half zero,all ones,all-equal" 2>&1 | grep -v amdgpu > $O/${T}_skew_phases.txt
cat $O/${T}_skew_phases.txt
