#!/bin/bash
# round 5, job o: the generic MSM at 2^24: phases and kernel timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05o}
mkdir -p $O
cd $R
python tools/timing/generic_phases.py 20 22 24 2>&1 | grep -v amdgpu > $O/${T}_generic_phases.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 $R/tools/timing/generic_phases.py 24 > /tmp/o.txt 2>&1
python3 $R/tools/timing/trace_summary.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) --tail 40 > $O/${T}_generic24_trace.txt 2>&1
cd $R
cat $O/${T}_generic_phases.txt; cat $O/${T}_generic24_trace.txt | cut -c1-150
