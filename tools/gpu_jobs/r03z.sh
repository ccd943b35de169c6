#!/bin/bash
# round 3, job z: Shoup products with scalar twiddles in the wave-uniform NTT stage pairs (BN254 Fr): suite, then A/B on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r03z_pytest.log 2>&1
grep -E "passed|failed|error" $O/r03z_pytest.log | tail -3
for rep in 1 2; do
  for f in 0 1; do
    echo "== MZK_NTT_SHOUP=$f (run $rep)" | tee -a $O/r03z_ntt_shoup_ab.txt
    MZK_NTT_SHOUP=$f python tools/timing/time_ntt.py 16,18,20,22,24 2>&1 | grep -v amdgpu.ids | grep Fr | tee -a $O/r03z_ntt_shoup_ab.txt
  done
done
python tools/timing/ntt_batch.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/r03z_ntt_shoup_ab.txt
