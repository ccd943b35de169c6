#!/bin/bash
# round 3, job w: sortless commit path up to 2^14 coefficients at 10-bit tables (new default widths): suite + small sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/r03w_pytest.log 2>&1
grep -E "passed|failed|error" $O/r03w_pytest.log | tail -3
python tools/timing/small_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/r03w_small_latency.txt
