#!/bin/bash
# round 6, job k: why is a mixed addition 12 % slower at 2^24 than at 2^20?  SQ counters of k_seg_accumulate at both sizes (separate runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
for lg in 20 24; do
  reps=20; [ $lg = 24 ] && reps=4
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/r06k_sq_$lg -- python3 $R/tools/timing/commit_only.py $lg $reps > $O/r06k_sq_$lg.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/r06k_tcc_$lg -- python3 $R/tools/timing/commit_only.py $lg $reps > $O/r06k_tcc_$lg.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/r06k_grbm_$lg -- python3 $R/tools/timing/commit_only.py $lg $reps > $O/r06k_grbm_$lg.log 2>&1
done
{
for lg in 20 24; do
  echo "== commit 2^$lg: SQ"; python3 $R/tools/timing/pmc_sq_summary.py $O/r06k_sq_$lg k_seg_accumulate
  echo "== commit 2^$lg: TCC"; python3 $R/tools/timing/pmc_sq_summary.py $O/r06k_tcc_$lg k_seg_accumulate
  echo "== commit 2^$lg: GRBM"; python3 $R/tools/timing/pmc_sq_summary.py $O/r06k_grbm_$lg k_seg_accumulate
  python3 - <<PY
import csv, glob
f = glob.glob("$O/r06k_sq_$lg/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "k_seg_accumulate" in r["Kernel_Name"]]
print("k_seg_accumulate durations (us): n=%d mean=%.1f min=%.1f" % (len(d), sum(d) / len(d), min(d)))
PY
done
} > $O/r06k_accumulate_2p20_vs_2p24_counters.txt 2>&1
find $O -path "*r06k*" -name "*.csv" -size +2M -delete
cat $O/r06k_accumulate_2p20_vs_2p24_counters.txt; tail -3 $O/r06k_tcc_24.log
