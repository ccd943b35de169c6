#!/bin/bash
# round 5, job q: two same-box A/Bs: (1) BN254 transforms with a straight-line radix-4 group for waves without unit twiddles (libmzk_hip_fru.so),
# (2) k_seg_combine with the next partial requested one addition ahead (libmzk_hip.so) against the build before it (libmzk_hip_nopf.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05q}
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_many.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
for rep in 1 2 3; do
for lib in libmzk_hip.so libmzk_hip_fru.so; do
  echo "== NTT $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_ntt.py 16,20,22,24 2>&1 | grep "^Fr" >> $O/${T}_ab.txt
done
done
MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_fru.so timeout 600 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q 2>&1 | tail -2 >> $O/${T}_ab.txt
for rep in 1 2; do
for lib in libmzk_hip_nopf.so libmzk_hip.so; do
  echo "== combine $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/commit_only.py 20 60 2>&1 | grep -v amdgpu | tail -1 >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/many_commit.py 10:256,12:64,14:16 2>&1 | grep -v amdgpu | cut -c1-60,150-260 >> $O/${T}_ab.txt
done
done
tail -3 $O/${T}_pytest.log; cat $O/${T}_ab.txt
