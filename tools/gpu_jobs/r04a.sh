#!/bin/bash
# round 3, job 4a: fused NTT edges for odd level counts too: NTT / poly / FRI tests, timings with and without
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r04a_pytest.log 2>&1
grep -E "passed|failed|error" $O/r04a_pytest.log | tail -3
for f in 0 1 0 1; do
  echo "== MZK_NTT_FUSE_EDGES=$f" | tee -a $O/r04a_ntt_odd_levels.txt
  MZK_NTT_FUSE_EDGES=$f python tools/timing/time_ntt.py 17,19,21,22,23 2>&1 | grep -v amdgpu.ids | tee -a $O/r04a_ntt_odd_levels.txt
done
