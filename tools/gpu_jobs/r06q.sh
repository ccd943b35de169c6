#!/bin/bash
# round 6, job q: where the 2.6 ms between commit_local and open_local of the end-to-end leg go -- the quotient pass alone, under the kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python tools/timing/open_quotient_time.py 22 2>&1 | grep -v amdgpu.ids | tee $O/r06q_open_quotient.txt
python tools/timing/open_quotient_time.py 20 2>&1 | grep -v amdgpu.ids | tee -a $O/r06q_open_quotient.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06q_prof -- python3 $R/tools/timing/open_quotient_time.py 22 > $O/r06q_prof.log 2>&1
find $O/r06q_prof -name "*kernel_stats.csv" -exec cp {} $O/r06q_kernel_stats.csv \;
head -12 $O/r06q_kernel_stats.csv | cut -c1-220
find $O/r06q_prof -name "*.csv" -size +2M -delete
