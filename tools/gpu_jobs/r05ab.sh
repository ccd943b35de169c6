#!/bin/bash
# round 5, job ab: device-side segment length: MSM tests, short / skewed scalars by phase, uniform commits
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ab}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_msm.py tests/test_gpu_many.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/skew_msm.py "uniform,16-bit scalars,64-bit,248-bit" 2>&1 | grep -v amdgpu > $O/${T}_skew_phases.txt
for lg in 20 24 16; do python tools/timing/commit_only.py $lg 40 2>&1 | grep -v amdgpu >> $O/${T}_skew_phases.txt; done
python tools/timing/many_commit.py 10:256,10:256:1:0:248,12:64,14:16,12:64:1:0:248,8:1024:1:0:248 2>&1 | grep -v amdgpu | cut -c1-100,180-400 >> $O/${T}_skew_phases.txt
tail -5 $O/${T}_pytest.log; cat $O/${T}_skew_phases.txt
