#!/bin/bash
# round 2, job a: full GPU suite (incl. the new full-size and multi-device tests), accumulate A/B sweep, SQ counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/r02a_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r02a_pytest.log
timeout 900 python tools/timing/acc_sweep.py 20 > $O/r02a_sweep20.log 2>&1
cd /tmp && export TMPDIR=/tmp
for pf in 0 1; do
  export MZK_ACC_PREFETCH=$pf
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
     --output-format csv -d $O/r02a_pmc_sq_pf$pf -- python3 $R/tools/timing/acc_sweep.py --child 20 > $O/r02a_pmc_sq_pf$pf.log 2>&1
  python3 $R/tools/timing/pmc_sq_summary.py $O/r02a_pmc_sq_pf$pf k_seg > $O/r02a_pmc_sq_pf$pf.txt 2>&1
  find $O/r02a_pmc_sq_pf$pf -name "*.csv" -size +3M -delete
done
unset MZK_ACC_PREFETCH
tail -5 $O/r02a_pytest.log; cat $O/r02a_sweep20.log; cat $O/r02a_pmc_sq_pf0.txt $O/r02a_pmc_sq_pf1.txt
