#!/bin/bash
# round 2, job z: SQ counters of the final tree's kernels -- the 2^20 NTT passes (both fields, default geometry) and the 2^20-pair
# commit's accumulate / combine (plain rocprofv3 --pmc runs, one counter set per run)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -f $O/r02z_sq.txt
for fid in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS \
     --output-format csv -d $O/r02z_pmc_ntt$fid -- python3 $R/tools/timing/ntt_only.py $fid 20 > $O/r02z_pmc_ntt$fid.log 2>&1
  echo "== NTT 2^20, field $fid (0 = Fr, 1 = M128)" >> $O/r02z_sq.txt
  python3 $R/tools/timing/pmc_sq_summary.py $O/r02z_pmc_ntt$fid k_ntt >> $O/r02z_sq.txt 2>&1
  find $O/r02z_pmc_ntt$fid -name "*.csv" -size +3M -delete
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS \
   --output-format csv -d $O/r02z_pmc_msm -- python3 $R/tools/timing/acc_sweep.py --child 20 > $O/r02z_pmc_msm.log 2>&1
echo "== 2^20-pair commits and generic MSMs" >> $O/r02z_sq.txt
python3 $R/tools/timing/pmc_sq_summary.py $O/r02z_pmc_msm k_seg >> $O/r02z_sq.txt 2>&1
find $O/r02z_pmc_msm -name "*.csv" -size +3M -delete
cat $O/r02z_sq.txt
