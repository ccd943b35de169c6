#!/bin/bash
# round 5, job aj: BN254 2^20 transform as two 512-lane workgroups per CU over 2048-element tiles (GeoF, -DMZK_NTT_FR_TWO_WG=1) against the shipped 1024-lane form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05aj}
mkdir -p $O
cd $R
rm -f $O/${T}_ab.txt
for rep in 1 2 3; do
for lib in libmzk_hip.so libmzk_hip_fr2.so libmzk_hip_fr2s1.so libmzk_hip_fr2s3.so; do
  [ -f myzkp_amd/$lib ] || continue
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_ntt.py 20 2>&1 | grep -v amdgpu | grep "^Fr" >> $O/${T}_ab.txt
done
done
cat $O/${T}_ab.txt
