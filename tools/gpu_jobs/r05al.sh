#!/bin/bash
# round 5, job al: several small Merkle levels per launch: tests (timing A/B against the previous library: see profiles/round5_merkle_levels_per_launch_ab.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05al}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri_protocol.py tests/test_gpu_dev_api.py tests/test_gpu_fuzz_slice.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
tail -6 $O/${T}_pytest.log
