#!/bin/bash
# round 5, job s: SQ counters of the Merkle hashing kernels (why the throughput levels run at ~45 % of the full-rate VALU expectation)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05s}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_SQ -- python3 $R/tools/timing/time_merkle.py > $O/${T}_SQ.log 2>&1
python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_SQ k_merkle > $O/${T}_sq_counters.txt 2>&1
cd $R
find $O -name "*.csv" -size +4M -delete
cat $O/${T}_sq_counters.txt | cut -c1-150
