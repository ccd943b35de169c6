#!/bin/bash
# round 3, job j: signed madd + one-multiply zero filter in the accumulate kernel: full GPU suite, commit timing, default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/r03j_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03j_pytest.log
tail -12 $O/r03j_pytest.log
timeout 600 python tools/timing/window_sweep.py 20,22,24 17 > $O/r03j_commit.txt 2>&1
timeout 600 python tools/timing/small_latency.py 4,10,12,14,16,18,20 >> $O/r03j_commit.txt 2>&1
timeout 600 python tools/timing/pcie_incl.py >> $O/r03j_commit.txt 2>&1
grep -v amdgpu.ids $O/r03j_commit.txt
timeout 1200 python bench.py > $O/r03j_bench.json 2> $O/r03j_bench.err
echo "bench rc=$?"; head -c 300 $O/r03j_bench.json; echo
