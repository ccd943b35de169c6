#!/bin/bash
# round 3, job q: the single-workgroup tail with LDS-resident steps and four Horner chains: suite, phases, and a sweep of the
# two thresholds (MZK_TAIL_MAX_OPS: widest step inside the tail; MZK_TAIL_ROW_MAX: widest step on row operations)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r03q_pytest.log 2>&1
grep -E "passed|failed|error" $O/r03q_pytest.log | tail -3
MZK_HIP_LIB=$R/scratch_whatif/trace/libmzk_hip.so python tools/timing/tail_trace.py 10 12 20 2>&1 | grep -v amdgpu.ids | tee $O/r03q_tail_trace.txt
for ops in 256 128 64; do for rm in 16 12 8 4; do
  echo "== MZK_TAIL_MAX_OPS=$ops MZK_TAIL_ROW_MAX=$rm" | tee -a $O/r03q_sweep.txt
  MZK_TAIL_MAX_OPS=$ops MZK_TAIL_ROW_MAX=$rm python tools/timing/small_latency.py 2>&1 | grep -E "2\^(10|12|14) " | tee -a $O/r03q_sweep.txt
  MZK_TAIL_MAX_OPS=$ops MZK_TAIL_ROW_MAX=$rm python tools/timing/commit_only.py 20 60 2>&1 | grep commit | tee -a $O/r03q_sweep.txt
done; done
