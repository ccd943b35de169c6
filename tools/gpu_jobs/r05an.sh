#!/bin/bash
# round 5, job an: small results through a pinned landing zone (d2h_sync): the whole GPU suite, then FRI / small-commit / pipeline timing against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05an}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
rm -f $O/${T}_ab.txt
for rep in 1 2; do
for lib in libmzk_hip_prev.so libmzk_hip.so; do
  [ -f myzkp_amd/$lib ] || continue
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/fri_round_cost.py 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/small_latency.py 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_merkle.py 2>&1 | grep -v amdgpu | grep "2^16\|2^20" >> $O/${T}_ab.txt
done
done
tail -4 $O/${T}_pytest.log; cat $O/${T}_ab.txt
