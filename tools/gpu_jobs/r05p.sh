#!/bin/bash
# round 5, job p: coarse bins of the generic layout's sort at 2^22 / 2^24: tuning build, MZK_COARSE_LOG_GENERIC = 8 | 9 | 10
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05p}
mkdir -p $O
cd $R
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
for rep in 1 2; do
for cl in 8 9 10; do
  echo "== MZK_COARSE_LOG_GENERIC=$cl (rep $rep)" >> $O/${T}_ab.txt
  MZK_COARSE_LOG_GENERIC=$cl python tools/timing/generic_phases.py 22 24 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
done
cat $O/${T}_ab.txt | cut -c1-200
