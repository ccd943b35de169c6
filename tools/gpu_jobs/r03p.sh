#!/bin/bash
# round 3, job p: last tree levels of k_small_accumulate on row additions (suite + small-size latency), and where the
# single-workgroup tail spends its time (a -DMZK_TAIL_TRACE build stamps its phases)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r03p_pytest.log 2>&1
tail -3 $O/r03p_pytest.log
python tools/timing/small_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/r03p_small_latency.txt
MZK_HIP_LIB=$R/scratch_whatif/trace/libmzk_hip.so python tools/timing/tail_trace.py 10 12 16 20 2>&1 | grep -v amdgpu.ids | tee $O/r03p_tail_trace.txt
