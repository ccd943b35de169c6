#!/bin/bash
# round 5, job y: MSM tests after the heavy-list change (pairs), short-coefficient timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05y}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests/test_gpu_many.py tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_dev_api.py tests/test_gpu_next_rows.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/many_commit.py 10:256,10:256:1:0:248,8:1024:1:0:248 2>&1 | grep -v amdgpu | cut -c1-100,180-400 > $O/${T}_many.txt
python tools/timing/skew_msm.py 2>&1 | grep -v amdgpu | tail -12 >> $O/${T}_many.txt
tail -5 $O/${T}_pytest.log; cat $O/${T}_many.txt
