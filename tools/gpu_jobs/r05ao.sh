#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ao}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri_protocol.py tests/test_gpu_next_rows.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/fri_round_cost.py 2>&1 | grep -v amdgpu > $O/${T}_fri.txt
python tools/timing/fri_round_cost.py 2>&1 | grep -v amdgpu >> $O/${T}_fri.txt
tail -4 $O/${T}_pytest.log; cat $O/${T}_fri.txt
