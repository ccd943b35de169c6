#!/bin/bash
# round 2, job h: SQ counters of the NTT kernels (2^20, both fields, both tile geometries)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "0 99" "1 0" "1 99"; do
  set -- $cfg
  fid=$1; lg=$2
  export MZK_NTT_LARGE_FR=$lg MZK_NTT_LARGE_M128=$lg
  tag=f${fid}_large${lg}
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS \
     --output-format csv -d $O/r02h_pmc_$tag -- python3 $R/tools/timing/ntt_only.py $fid 20 > $O/r02h_pmc_$tag.log 2>&1
  echo "== field $fid, large tiles from 2^$lg" >> $O/r02h_ntt_sq.txt
  python3 $R/tools/timing/pmc_sq_summary.py $O/r02h_pmc_$tag k_ntt >> $O/r02h_ntt_sq.txt 2>&1
  find $O/r02h_pmc_$tag -name "*.csv" -size +3M -delete
done
cat $O/r02h_ntt_sq.txt
