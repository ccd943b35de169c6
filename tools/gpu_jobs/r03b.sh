#!/bin/bash
# round 3, job b: fixed-base window width by SRS size: commit time for c = 15..20 at 2^16..2^24 pairs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python tools/timing/window_sweep.py 16,17,18,19,20,21,22 14,15,16,17,18,19,20 > $O/r03b_window_sweep.txt 2>&1
timeout 900 python tools/timing/window_sweep.py 23,24 15,16,17,18,19 >> $O/r03b_window_sweep.txt 2>&1
cat $O/r03b_window_sweep.txt
