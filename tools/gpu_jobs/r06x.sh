#!/bin/bash
# round 6, job x: the leaf level of small trees by lane pairs: Merkle / FRI parity, then same-box A/B (tuning build, MZK_LEAF_LANE_PAIRS=0 = one lane per leaf pair)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri_protocol.py tests/test_gpu_next_rows.py tests/test_gpu_cpp_mirror.py tests/test_gpu_fuzz_slice.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r06x_pytest.txt
rm -f $O/r06x_ab.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== MZK_LEAF_LANE_PAIRS=$v (rep $rep)" >> $O/r06x_ab.txt
    MZK_LEAF_LANE_PAIRS=$v python tools/timing/time_merkle.py 2>&1 | grep -v amdgpu.ids | grep "2^16\|2^20" >> $O/r06x_ab.txt
    MZK_LEAF_LANE_PAIRS=$v python tools/timing/fri_round_cost.py 2>&1 | grep -v amdgpu.ids >> $O/r06x_ab.txt
  done
done
cat $O/r06x_ab.txt
