#!/bin/bash
# round 3, final job (the tree as committed last): full GPU suite, default bench line, kernel stats of the same bench, HBM traffic
# counters of the commit's kernels, small-size latency, and the N > 1 bench path rehearsed through the launcher on the one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r03final}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
timeout 1200 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "bench rc=$?" >> $O/${T}_pytest.log
python tools/timing/small_latency.py > $O/${T}_small_latency.txt 2>&1
python tools/timing/pcie_incl.py > $O/${T}_pcie_inclusive.txt 2>&1
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --steps 10 --warmup 2 --skip-cpu --extra-sizes= --e2e-log2n 0 --strong-log2n 0 --strong-ntt-log2n 0 --no-two-in-flight"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $BENCH > $O/${T}_bench_under_rocprof.json 2> $O/${T}_prof.err
find $O/${T}_prof -name "*kernel_stats.csv" -exec cp {} $O/${T}_bench_kernel_stats.csv \;
python3 $R/tools/timing/prof_split.py $(find $O/${T}_prof -name "*kernel_trace.csv" | head -1) > $O/${T}_per_msm_kernel_budget.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_msm_$c -- python3 $R/tools/timing/commit_only.py 20 40 > $O/${T}_msm_$c.log 2>&1
done
{ echo "== KZG commit 2^20, default window width (commit_only.py 20 40)"; python3 $R/tools/timing/pmc_summary.py $O/${T}_msm_FETCH_SIZE $O/${T}_msm_WRITE_SIZE; } > $O/${T}_hbm_traffic_pmc.txt 2>&1
cd $R
export MZK_BENCH_SHARED_GPU_TEST=1 MZK_BENCH_WATCHDOG_S=500
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --log2n 18 --extra-sizes= --e2e-log2n 18 --strong-log2n 20 --strong-ntt-log2n 20 > $O/${T}_rehearsal_world2.json 2> $O/${T}_rehearsal_world2.err
echo "rehearsal world 2 rc=$?" >> $O/${T}_pytest.log
timeout 900 python bench.py --gpus 4 --steps 3 --warmup 1 --log2n 18 --extra-sizes= --e2e-log2n 18 --strong-log2n 20 --strong-ntt-log2n 20 > $O/${T}_rehearsal_world4.json 2> $O/${T}_rehearsal_world4.err
echo "rehearsal world 4 rc=$?" >> $O/${T}_pytest.log
find $O -name "*.csv" -size +4M -delete
tail -16 $O/${T}_pytest.log; head -c 500 $O/${T}_bench.json; echo; cat $O/${T}_per_msm_kernel_budget.txt | head -45; head -6 $O/${T}_hbm_traffic_pmc.txt; grep -v amdgpu $O/${T}_small_latency.txt; grep -v amdgpu $O/${T}_pcie_inclusive.txt
