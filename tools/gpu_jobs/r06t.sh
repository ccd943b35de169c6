#!/bin/bash
# round 6, job t: the Keccak round constant fetched one round ahead in the lane-pair hash (the scalar load + wait sat in front of every round of the
# lone wave of a tree's upper levels): Merkle / FRI parity, then same-box A/B against a library with the fetch inside the round
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_rc_unrolled.so timeout 1500 python -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri_protocol.py tests/test_gpu_next_rows.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r06t_pytest.txt
rm -f $O/r06t_ab.txt
for rep in 1 2; do
  for lib in libmzk_hip_rc_in_round.so libmzk_hip.so libmzk_hip_rc_unrolled.so; do
    echo "== $lib (rep $rep)" >> $O/r06t_ab.txt
    MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_merkle.py 2>&1 | grep -v amdgpu.ids >> $O/r06t_ab.txt
    MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/fri_round_cost.py 2>&1 | grep -v amdgpu.ids >> $O/r06t_ab.txt
  done
done
cat $O/r06t_ab.txt
