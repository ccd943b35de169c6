#!/bin/bash
# round 2, job u: sizes beyond BASELINE's 2^24 up to the ABI's stated limits, closed-form checked (tools/timing/max_sizes.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
free -g | head -2 > $O/r02u_max_sizes.txt
timeout 600 python tools/timing/max_sizes.py ntt 20 msm 20 >> $O/r02u_max_sizes.txt 2>&1
echo "small rc=$?" >> $O/r02u_max_sizes.txt
timeout 1500 python tools/timing/max_sizes.py ntt 26 28 >> $O/r02u_max_sizes.txt 2>&1
echo "ntt rc=$?" >> $O/r02u_max_sizes.txt
timeout 1500 python tools/timing/max_sizes.py msm 26 27 >> $O/r02u_max_sizes.txt 2>&1
echo "msm rc=$?" >> $O/r02u_max_sizes.txt
cat $O/r02u_max_sizes.txt
