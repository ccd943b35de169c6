#!/bin/bash
# round 5, job w: per-kernel times of the grid-batched commit, full-width against 248-bit coefficients
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05w}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in 10:256 10:256:1:0:248; do
  tag=$(echo $spec | tr ':' '_')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_$tag -- python3 $R/tools/timing/many_commit.py $spec > $O/${T}_$tag.log 2>&1
  find $O/${T}_$tag -name "*kernel_stats.csv" -exec cp {} $O/${T}_${tag}_kernel_stats.csv \;
  echo "== $spec"; head -12 $O/${T}_${tag}_kernel_stats.csv | cut -c1-150
done
find $O -name "*.csv" -size +4M -delete
