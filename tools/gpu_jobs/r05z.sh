#!/bin/bash
# round 5, job z: the shared-bucket form of k_seg_combine_heavy: MSM tests, skewed-input timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05z}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_msm.py tests/test_gpu_many.py -m gpu -x -q -k "msm or commit or skew or heavy or many" ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/skew_msm.py 2>&1 | grep -v amdgpu | tail -12 > $O/${T}_skew.txt
python tools/timing/commit_only.py 20 40 2>&1 | grep -v amdgpu >> $O/${T}_skew.txt
python tools/timing/many_commit.py 10:256,10:256:1:0:248 2>&1 | grep -v amdgpu | cut -c1-100,180-400 >> $O/${T}_skew.txt
tail -5 $O/${T}_pytest.log; cat $O/${T}_skew.txt
