#!/bin/bash
# round 6, final job (the tree as committed): full GPU suite, default bench line, kernel stats of the same bench, hardware counters of the
# dominant kernels ON THIS TREE -- SQ counters and FETCH_SIZE / WRITE_SIZE of the Fr and M128 transforms, of the KZG commit and of the
# GENERIC MSM (msm_generic.roofline.traffic) --, small-size latency, the many-commit shapes, the STARK commit pipeline, FRI round costs.
# Every rocprofv3 run has the program directly after `--`, counters in runs of their own.  The counter summaries carry the fingerprint of
# the kernel sources they were taken from (tools/source_fingerprint.py: bench.py's traffic_stale).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r06final}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
timeout 1200 python bench.py --detail-file $O/${T}_bench_default_detail.json > $O/${T}_bench_default.json 2> $O/${T}_bench.err
echo "bench rc=$?" >> $O/${T}_pytest.log
python tools/timing/small_latency.py > $O/${T}_small_latency.txt 2>&1
python tools/timing/pcie_incl.py > $O/${T}_pcie_inclusive.txt 2>&1
python tools/timing/many_commit.py 10:256,10:256:1:10,10:256:1:12,12:64,12:64:1:10,8:1024,8:1024:1:10,13:32,14:16,10:256:1:0:248,12:64:1:0:248,8:1024:1:0:248 > $O/${T}_many_commit.txt 2>&1
python tools/timing/skew_msm.py "uniform,bits,bytes,16-bit scalars,32-bit,64-bit,128-bit,248-bit,half zero,all ones,all-equal" > $O/${T}_short_scalars.txt 2>&1
python tools/timing/stark_commit_pipeline.py 12 16 > $O/${T}_stark_commit_pipeline.txt 2>&1
python tools/timing/stark_commit_pipeline.py 14 16 >> $O/${T}_stark_commit_pipeline.txt 2>&1
python tools/timing/fri_round_cost.py > $O/${T}_fri_round_cost.txt 2>&1
python tools/timing/time_ntt.py 10,14,16,18,20,21,22,24 > $O/${T}_time_ntt.txt 2>&1
python tools/timing/generic_phases.py 20 22 23 24 > $O/${T}_generic_phases.txt 2>&1
python tools/timing/window_sweep.py 16,18,20,22,24 1 2>&1 | cut -c1-230 > $O/${T}_commit_phases.txt
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --steps 10 --warmup 2 --skip-cpu --extra-sizes= --e2e-log2n 0 --strong-log2n 0 --strong-ntt-log2n 0 --no-two-in-flight --detail-file $O/${T}_bench_under_rocprof_detail.json"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $BENCH > $O/${T}_bench_under_rocprof.json 2> $O/${T}_prof.err
find $O/${T}_prof -name "*kernel_stats.csv" -exec cp {} $O/${T}_bench_kernel_stats.csv \;
python3 $R/tools/timing/prof_split.py $(find $O/${T}_prof -name "*kernel_trace.csv" | head -1) > $O/${T}_per_msm_kernel_budget.txt 2>&1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"
run_pmc() {   # tag, program and arguments...
  local tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_${tag}_SQ -- python3 "$@" > $O/${T}_${tag}_SQ.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_${tag}_$c -- python3 "$@" > $O/${T}_${tag}_$c.log 2>&1
  done
}
run_pmc nttfr $R/tools/timing/ntt_only.py 0 20
run_pmc nttm128 $R/tools/timing/ntt_only.py 1 20
run_pmc commit $R/tools/timing/commit_only.py 20 40
run_pmc generic $R/tools/timing/generic_phases.py 20
{
  python3 $R/tools/source_fingerprint.py
  echo "== KZG commit 2^20, default window width (commit_only.py 20 40)"; python3 $R/tools/timing/pmc_summary.py $O/${T}_commit_FETCH_SIZE $O/${T}_commit_WRITE_SIZE
  echo "== NTT Fr 2^20 (ntt_only.py 0 20)"; python3 $R/tools/timing/pmc_summary.py $O/${T}_nttfr_FETCH_SIZE $O/${T}_nttfr_WRITE_SIZE | grep k_ntt
  echo "== NTT M128 2^20 (ntt_only.py 1 20)"; python3 $R/tools/timing/pmc_summary.py $O/${T}_nttm128_FETCH_SIZE $O/${T}_nttm128_WRITE_SIZE | grep k_ntt
  echo "== generic MSM 2^20 (generic_phases.py 20)"; python3 $R/tools/timing/pmc_summary.py $O/${T}_generic_FETCH_SIZE $O/${T}_generic_WRITE_SIZE
} > $O/${T}_hbm_traffic_pmc.txt 2>&1
{
  python3 $R/tools/source_fingerprint.py
  echo "== NTT Fr 2^20"; python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_nttfr_SQ k_ntt
  echo "== NTT M128 2^20"; python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_nttm128_SQ k_ntt
  echo "== KZG commit 2^20"; python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_commit_SQ k_seg_accumulate k_seg_combine k_fine_scatter k_coarse_scatter k_reduce_tail k_halve_multi
  echo "== generic MSM 2^20"; python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_generic_SQ k_seg_accumulate
} > $O/${T}_sq_counters.txt 2>&1
cd $R
python3 tools/kernel_resources.py --priced > $O/${T}_kernel_resources.txt 2>&1
find $O -name "*.csv" -size +4M -delete
tail -16 $O/${T}_pytest.log; head -c 600 $O/${T}_bench_default.json; echo; head -30 $O/${T}_per_msm_kernel_budget.txt; cat $O/${T}_hbm_traffic_pmc.txt | cut -c1-160 | head -50; cat $O/${T}_sq_counters.txt | head -90; grep -v amdgpu $O/${T}_small_latency.txt; grep -v amdgpu $O/${T}_many_commit.txt | cut -c1-200; grep -v amdgpu $O/${T}_stark_commit_pipeline.txt; grep -v amdgpu $O/${T}_fri_round_cost.txt; grep -v amdgpu $O/${T}_time_ntt.txt; grep -v amdgpu $O/${T}_generic_phases.txt | cut -c1-220; grep -v amdgpu $O/${T}_short_scalars.txt
