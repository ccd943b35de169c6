#!/bin/bash
# round 3, job e: kernel-by-kernel trace of small commits and of a 2^20 commit with the row tails (and the quad tails beside them)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  export MZK_ROW_TAILS=$f
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03e_small_$f -- python3 $R/tools/timing/small_trace.py > $O/r03e_small_$f.log 2>&1
  echo "== small commits, MZK_ROW_TAILS=$f" >> $O/r03e_trace.txt
  python3 $R/tools/timing/small_trace_summary.py $O/r03e_small_$f >> $O/r03e_trace.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03e_big_$f -- python3 $R/tools/timing/commit_only.py 20 30 > $O/r03e_big_$f.log 2>&1
  echo "== 2^20 commit, MZK_ROW_TAILS=$f" >> $O/r03e_trace.txt
  python3 $R/tools/timing/prof_split.py $(find $O/r03e_big_$f -name "*kernel_trace.csv" | head -1) >> $O/r03e_trace.txt 2>&1
done
find $O -path "*r03e_*" -name "*.csv" -size +3M -delete
cat $O/r03e_trace.txt
