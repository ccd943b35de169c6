#!/bin/bash
# round 3, job x: NTT passes with the first / last stage pair on registers next to the global loads / stores: suite + timings
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r03x_pytest.log 2>&1
grep -E "passed|failed|error" $O/r03x_pytest.log | tail -3
python tools/timing/time_ntt.py 2>&1 | grep -v amdgpu.ids | tee $O/r03x_time_ntt.txt
python tools/timing/ntt_batch.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/r03x_ntt_batch.txt
