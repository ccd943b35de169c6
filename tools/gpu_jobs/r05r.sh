#!/bin/bash
# round 5, job r: 2^20 transforms on the small-tile geometry (three passes of 7 + 7 + 6 levels, four workgroups per CU) against the large tiles
# (two passes of 2^10 levels, one workgroup per CU): tuning build, MZK_NTT_LARGE_FR / MZK_NTT_LARGE_M128 = 99 (never large) | default
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05r}
mkdir -p $O
cd $R
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
for rep in 1 2; do
  echo "== default geometry (rep $rep)" >> $O/${T}_ab.txt
  python tools/timing/time_ntt.py 20,21,22 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
  echo "== small tiles forced: MZK_NTT_LARGE_FR=99 MZK_NTT_LARGE_M128=99 (rep $rep)" >> $O/${T}_ab.txt
  MZK_NTT_LARGE_FR=99 MZK_NTT_LARGE_M128=99 python tools/timing/time_ntt.py 20,21,22 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
  echo "== large tiles forced from 2^20: MZK_NTT_LARGE_FR=20 MZK_NTT_LARGE_M128=20 (rep $rep)" >> $O/${T}_ab.txt
  MZK_NTT_LARGE_FR=20 MZK_NTT_LARGE_M128=20 python tools/timing/time_ntt.py 20,21,22 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
cat $O/${T}_ab.txt
