#!/bin/bash
# round 3, job 4g: what-if -- the coarse scatter with perfectly coalesced (wrong) stores: how much of its 72 us is the scatter?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in shipped whatif; do
  if [ $lib = whatif ]; then export MZK_HIP_LIB=$R/scratch_whatif/coal/libmzk_hip.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04g_$lib -- python3 $R/tools/timing/commit_only.py 20 20 > $O/r04g_$lib.log 2>&1
  echo "== $lib"; grep -h "k_coarse\|k_fine\|k_seg_acc" $(find $O/r04g_$lib -name "*kernel_stats.csv") | cut -c1-60,100-200 | head
done 2>&1 | tee $O/r04g_coalesced_whatif.txt
find $O -name "*.csv" -size +2M -delete
