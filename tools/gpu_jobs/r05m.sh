#!/bin/bash
# round 5, job m: polynomial-tree tests after the shared-denominator division, the STARK pipeline leg of the bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05m}
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_poly.py tests/test_gpu_coset_divide.py tests/test_gpu_next_rows.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/stark_commit_pipeline.py 14 16 > $O/${T}_stark_pipeline_2pow14.txt 2>&1
python tools/timing/stark_commit_pipeline.py 12 16 > $O/${T}_stark_pipeline_2pow12.txt 2>&1
tail -3 $O/${T}_pytest.log; grep -v amdgpu $O/${T}_stark_pipeline_2pow14.txt; grep -v amdgpu $O/${T}_stark_pipeline_2pow12.txt
