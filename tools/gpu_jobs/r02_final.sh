#!/bin/bash
# round 2, job r02final (end of round 2: the tree as committed last): full GPU suite, default bench line, kernel stats of the same bench, small-size latency
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02final_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r02final_pytest.log
timeout 1200 python bench.py > $O/r02final_bench.json 2> $O/r02final_bench.err
python tools/timing/small_latency.py > $O/r02final_small_latency.txt 2>&1
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --steps 10 --warmup 2 --skip-cpu --extra-sizes= --e2e-log2n 0 --strong-log2n 0 --no-two-in-flight"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02final_prof -- python3 $BENCH > $O/r02final_bench_under_rocprof.json 2> $O/r02final_prof.err
find $O/r02final_prof -name "*kernel_stats.csv" -exec cp {} $O/r02final_bench_kernel_stats.csv \;
python3 $R/tools/timing/prof_split.py $(find $O/r02final_prof -name "*kernel_trace.csv" | head -1) > $O/r02final_per_msm_kernel_budget.txt 2>&1
find $O -name "*.csv" -size +4M -delete
tail -4 $O/r02final_pytest.log; head -c 600 $O/r02final_bench.json; echo; cat $O/r02final_per_msm_kernel_budget.txt; cat $O/r02final_small_latency.txt
