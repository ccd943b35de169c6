#!/bin/bash
# round 2, job i: LDS / wait counters of the 2^20 Fr NTT kernels (large tiles)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -f $O/r02i_ntt_lds.txt
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/r02i_pmc_$tag -- python3 $R/tools/timing/ntt_only.py 0 20 > $O/r02i_pmc_$tag.log 2>&1
  python3 $R/tools/timing/pmc_sq_summary.py $O/r02i_pmc_$tag k_ntt >> $O/r02i_ntt_lds.txt 2>&1
  tail -3 $O/r02i_pmc_$tag.log >> $O/r02i_ntt_lds.txt
  find $O/r02i_pmc_$tag -name "*.csv" -size +3M -delete
done
cat $O/r02i_ntt_lds.txt
