#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ae}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests/test_gpu_msm.py -m gpu -x -q --durations=5 ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
tail -12 $O/${T}_pytest.log
