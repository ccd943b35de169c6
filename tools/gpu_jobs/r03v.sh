#!/bin/bash
# round 3, job v: how far up does the sortless small-commit path pay?  commits of 2^12..2^16 against 10..13-bit tables with the
# path's size limit raised (MZK_SCAN_MAX_LOG), against the default widths through the general pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== MZK_SCAN_MAX_LOG=16" | tee $O/r03v_scan_sweep.txt
MZK_SCAN_MAX_LOG=16 python tools/timing/window_sweep.py 12,13,14,15,16 10,11,12,13 2>&1 | grep -v amdgpu.ids | tee -a $O/r03v_scan_sweep.txt
echo "== default path (general pipeline above 4096), 13 / 16 bits" | tee -a $O/r03v_scan_sweep.txt
python tools/timing/window_sweep.py 13,14,15,16 13,16 2>&1 | grep -v amdgpu.ids | tee -a $O/r03v_scan_sweep.txt
