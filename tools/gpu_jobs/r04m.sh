#!/bin/bash
# round 3, job 4m: records per fine-sort workgroup (MZK_PER_FINE): 32768 (shipped) vs 16384 vs 65536
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do for pf in 32768 16384 65536 8192; do
  echo "== MZK_PER_FINE=$pf (run $rep)" | tee -a $O/r04m_per_fine_ab.txt
  MZK_PER_FINE=$pf python tools/timing/window_sweep.py 18,20,22 1 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tee -a $O/r04m_per_fine_ab.txt
  MZK_PER_FINE=$pf python tools/timing/generic_phases.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tee -a $O/r04m_per_fine_ab.txt
done; done
