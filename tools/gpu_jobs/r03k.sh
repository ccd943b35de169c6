#!/bin/bash
# round 3, job k: A/B of the inter-pass twiddle L2 warm-up in k_ntt_strided (shipped build vs -DMZK_NTT_TW_WARM=0), interleaved runs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2 3; do
  echo "== shipped (warm-up on), run $rep" >> $O/r03k_ntt_warm.txt
  timeout 300 python tools/timing/time_ntt.py 16,20,22,24 >> $O/r03k_ntt_warm.txt 2>&1
  echo "== -DMZK_NTT_TW_WARM=0, run $rep" >> $O/r03k_ntt_warm.txt
  MZK_HIP_LIB=$R/scratch_whatif/nowarm/libmzk_hip.so timeout 300 python tools/timing/time_ntt.py 16,20,22,24 >> $O/r03k_ntt_warm.txt 2>&1
done
grep -v amdgpu.ids $O/r03k_ntt_warm.txt
