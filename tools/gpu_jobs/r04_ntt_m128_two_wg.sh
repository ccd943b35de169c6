#!/bin/bash
# round 4: M128 large tiles as two 512-lane workgroups per CU (twiddles from global / scalar memory) against round 3's single
# 1024-lane workgroup, same box, tuning build; then the shipped library; then the NTT tests.
# MZK_NTT_M128_TWO_WG = smallest log2 size that takes the two-workgroup geometry (99 = never, 20 = from 2^20 on).  The record
# profiles/round4_ntt_m128_two_workgroups_ab.txt was taken while the switch was still a boolean {0, 1}; with the committed meaning both of
# those values select the two-workgroup form everywhere -- the loop below uses the values that reproduce the A/B.
O=gpurun_out; mkdir -p $O
T=$PWD/myzkp_amd/libmzk_hip_tuning.so
for rep in 1 2; do
  for f in 99 20; do
    echo "== MZK_NTT_M128_TWO_WG=$f (run $rep)" | tee -a $O/r04_ntt_m128_two_wg.txt
    MZK_HIP_LIB=$T MZK_NTT_M128_TWO_WG=$f python tools/timing/time_ntt.py 20,25 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_ntt_m128_two_wg.txt
  done
done
echo "== shipped library" | tee -a $O/r04_ntt_m128_two_wg.txt
python tools/timing/time_ntt.py 16,18,20,21,22,24 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_ntt_m128_two_wg.txt
python tools/timing/time_lde.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O/r04_ntt_m128_two_wg.txt
python -m pytest tests/test_gpu_ntt.py tests/test_gpu_row_ec.py -x -q 2>&1 | tail -4
