#!/bin/bash
# round 2, job e (final state of the round): full GPU suite, default bench line, rocprofv3 kernel stats + HBM traffic counters of the same command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02e_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r02e_pytest.log
timeout 1200 python bench.py > $O/r02e_bench.json 2> $O/r02e_bench.err
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --steps 10 --warmup 2 --skip-cpu --extra-sizes= --e2e-log2n 0 --strong-log2n 0"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02e_prof -- python3 $BENCH > $O/r02e_bench_under_rocprof.json 2> $O/r02e_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r02e_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --skip-cpu --extra-sizes= --e2e-log2n 0 --strong-log2n 0 > /dev/null 2> $O/r02e_pmc_$c.err
done
python3 $R/tools/timing/pmc_summary.py $O/r02e_pmc_FETCH_SIZE $O/r02e_pmc_WRITE_SIZE > $O/r02e_hbm_traffic_pmc.txt 2>&1
find $O/r02e_prof -name "*kernel_stats.csv" -exec cp {} $O/r02e_bench_kernel_stats.csv \;
python3 $R/tools/timing/prof_split.py $(find $O/r02e_prof -name "*kernel_trace.csv" | head -1) > $O/r02e_per_msm_kernel_budget.txt 2>&1
find $O -name "*.csv" -size +4M -delete
cd $R && python tools/timing/small_latency.py > $O/r02e_small_latency.txt 2>&1; python tools/timing/time_ntt.py > $O/r02e_ntt_sizes.txt 2>&1; python tools/timing/time_poly.py > $O/r02e_poly_trees.txt 2>&1; python tools/timing/pipelined_commits.py 20 > $O/r02e_two_in_flight.txt 2>&1
tail -4 $O/r02e_pytest.log; head -c 1500 $O/r02e_bench.json; echo; cat $O/r02e_per_msm_kernel_budget.txt; cat $O/r02e_small_latency.txt; head -12 $O/r02e_hbm_traffic_pmc.txt
