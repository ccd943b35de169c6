#!/bin/bash
# round 3, job g: (1) cost of the final inversion (fold of one record), (2) segment combine one lane per bucket from 2^16 buckets,
# (3) latency table with the re-tuned row tails
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03g_fold -- python3 $R/tools/timing/fold_one.py > $O/r03g_fold.log 2>&1
grep "fold of" $O/r03g_fold.log > $O/r03g.txt
find $O/r03g_fold -name "*kernel_stats.csv" -exec grep -h "fold_partials" {} \; >> $O/r03g.txt
cd $R
for w in 17 16; do
  echo "== MZK_COMBINE_WIDE_MIN_LOG=$w" >> $O/r03g.txt
  MZK_COMBINE_WIDE_MIN_LOG=$w timeout 600 python tools/timing/window_sweep.py 20,22 17 >> $O/r03g.txt 2>&1
done
echo "== latency, row tails" >> $O/r03g.txt
timeout 600 python tools/timing/small_latency.py 4,10,12,13,14,16,18,20 >> $O/r03g.txt 2>&1
grep -v amdgpu.ids $O/r03g.txt
