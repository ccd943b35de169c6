#!/bin/bash
# round 5, job i: kernel traces of the STARK commit pipeline's stages (what the interpolation and a FRI round spend their time on)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05i}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in interp fri; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_$w -- python3 $R/tools/timing/stark_stage_trace.py $w 14 16 4 > $O/${T}_$w.log 2>&1
find $O/${T}_$w -name "*kernel_stats.csv" -exec cp {} $O/${T}_${w}_kernel_stats.csv \;
python3 $R/tools/timing/trace_summary.py $(find $O/${T}_$w -name "*kernel_trace.csv" | head -1) > $O/${T}_${w}_timeline.txt 2>&1
done
cd $R
find $O -name "*.csv" -size +4M -delete
for w in interp fri; do grep -v amdgpu $O/${T}_$w.log | tail -4; head -25 $O/${T}_${w}_kernel_stats.csv | cut -c1-140; done
