#!/bin/bash
# round 5, job ad: fine slices by bin size (fine_plan): MSM tests, then uniform / short scalars against the previous library, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ad}
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_msm.py tests/test_gpu_many.py tests/test_gpu_next_rows.py tests/test_gpu_multi.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
rm -f $O/${T}_ab.txt
for rep in 1 2; do
for lib in libmzk_hip_prev.so libmzk_hip.so; do
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  for lg in 20 24 16 18 14; do MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/commit_only.py $lg 40 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt; done
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/skew_msm.py "uniform,bytes,16-bit scalars,248-bit" 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/generic_phases.py 20 24 2>&1 | grep -v amdgpu | cut -c1-200 >> $O/${T}_ab.txt
done
done
tail -5 $O/${T}_pytest.log; cat $O/${T}_ab.txt
