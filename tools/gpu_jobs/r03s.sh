#!/bin/bash
# round 3, job s: tail with one inlined doubling / addition in the Horner loop (calls only for the one-shot sites): suite,
# phases, small sizes, 2^20 commit and generic MSM
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r03s_pytest.log 2>&1
grep -E "passed|failed|error" $O/r03s_pytest.log | tail -3
MZK_HIP_LIB=$R/scratch_whatif/trace/libmzk_hip.so python tools/timing/tail_trace.py 10 12 20 2>&1 | grep -v amdgpu.ids | tee $O/r03s_tail_trace.txt
python tools/timing/small_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/r03s_small_latency.txt
python tools/timing/commit_only.py 20 100 2>&1 | grep commit | tee -a $O/r03s_small_latency.txt
python tools/timing/time_msm.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/r03s_time_msm.txt
