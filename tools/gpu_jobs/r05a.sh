#!/bin/bash
# round 5, job a: the sparse-modulus M128 product and the signed lazy butterflies -- full GPU suite (fe_mul<M128> changed for every
# M128 kernel), transform timings, SQ counters of the M128 passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05a}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/time_ntt.py 18,20,21,22,24 > $O/${T}_time_ntt.txt 2>&1
python tools/timing/time_lde.py > $O/${T}_time_lde.txt 2>&1
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_nttm128_SQ -- python3 $R/tools/timing/ntt_only.py 1 20 > $O/${T}_nttm128_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_nttm128_trace -- python3 $R/tools/timing/ntt_only.py 1 20 > $O/${T}_nttm128_trace.log 2>&1
python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_nttm128_SQ k_ntt > $O/${T}_sq_counters.txt 2>&1
find $O/${T}_nttm128_trace -name "*kernel_stats.csv" -exec cp {} $O/${T}_nttm128_kernel_stats.csv \;
cd $R
find $O -name "*.csv" -size +4M -delete
tail -12 $O/${T}_pytest.log; grep -v amdgpu $O/${T}_time_ntt.txt; grep -v amdgpu $O/${T}_time_lde.txt | tail -12; cat $O/${T}_sq_counters.txt; head -5 $O/${T}_nttm128_kernel_stats.csv | cut -c1-200
