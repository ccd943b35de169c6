#!/bin/bash
# round 5, job h: same-box A/B of two or more builds of the library on the transforms: LIBS="a.so b.so" (names under myzkp_amd/), SIZES
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05h}
LIBS=${LIBS:-"libmzk_hip_nont.so libmzk_hip.so"}
SIZES=${SIZES:-"16,18,20,22,24"}
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
for rep in 1 2 3; do
for lib in $LIBS; do
  [ -f myzkp_amd/$lib ] || continue
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_ntt.py $SIZES 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
done
tail -3 $O/${T}_pytest.log; cat $O/${T}_ab.txt
