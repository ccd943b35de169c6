#!/bin/bash
# round 6, job s: the zerofier of a subgroup prefix in one launch: parity, a fuzz seed (the fuzzer now checks fast_zerofier too), the call's time with and without the path
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_poly.py tests/test_gpu_cpp_mirror.py -m gpu -x -q 2>&1 | tail -6 | tee $O/r06s_pytest.txt
bash tools/gpu_jobs/r06_fuzz.sh 6301 > /dev/null 2>&1
cp $O/r06_differential_fuzz.txt $O/r06s_fuzz.txt
{
  for rep in 1 2; do
    echo "== shipped library"; python tools/timing/zerofier_prefix_time.py 14 2>&1 | grep -v amdgpu.ids
    echo "== tuning build, MZK_INTERP_PREFIX=0 (subproduct tree)"; MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so MZK_INTERP_PREFIX=0 python tools/timing/zerofier_prefix_time.py 14 2>&1 | grep -v amdgpu.ids
  done
} > $O/r06s_zerofier_ab.txt
cat $O/r06s_fuzz.txt | sed "s/: .*; /: ... /"; cat $O/r06s_zerofier_ab.txt
