#!/bin/bash
# round 3, job 4l: fine scatter stage capacity 16384 (shipped) vs 8192 vs 4096 records per round (LDS 104 / 56 / 32 KB per
# workgroup: one, two, four workgroups per CU): sort phase of the commit and of the generic MSM
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do for cap in 16384 8192 4096; do
  if [ $cap = 16384 ]; then unset MZK_HIP_LIB; else export MZK_HIP_LIB=$R/scratch_whatif/cap$cap/libmzk_hip.so; fi
  echo "== STAGE_CAP=$cap (run $rep)" | tee -a $O/r04l_stage_cap_ab.txt
  python tools/timing/window_sweep.py 18,20,22 1 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a $O/r04l_stage_cap_ab.txt
  python tools/timing/generic_phases.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tee -a $O/r04l_stage_cap_ab.txt
done; done
