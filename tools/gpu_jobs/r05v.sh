#!/bin/bash
# round 5, job v: the grid-batched commit on SHORT coefficients (31-byte chunks, the reference's DAS callers) against full-width ones
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05v}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_many.py tests/test_gpu_msm.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/many_commit.py 10:256,10:256:1:0:248,12:64,12:64:1:0:248,8:1024,8:1024:1:0:248,14:16,14:16:1:0:248,10:256:1:10,10:256:1:10:248 2>&1 | grep -v amdgpu > $O/${T}_many_short.txt
tail -5 $O/${T}_pytest.log; cat $O/${T}_many_short.txt
