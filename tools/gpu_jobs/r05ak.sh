#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ak}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_poly.py -m gpu -x -q --durations=5 ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
tail -12 $O/${T}_pytest.log
