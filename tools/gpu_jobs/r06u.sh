#!/bin/bash
# round 6, job u: the root of a FRI round's tree posted by the tail kernel into mapped host memory (the host spins on a sequence number) instead of
# copy engine + stream synchronize: Merkle / FRI parity, then same-box A/B (tuning build, MZK_ROOT_MAILBOX=0 = copy + synchronize)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri_protocol.py tests/test_gpu_next_rows.py tests/test_gpu_multi.py tests/test_gpu_cpp_mirror.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r06u_pytest.txt
rm -f $O/r06u_ab.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== MZK_ROOT_MAILBOX=$v (rep $rep)" >> $O/r06u_ab.txt
    MZK_ROOT_MAILBOX=$v python tools/timing/time_merkle.py 2>&1 | grep -v amdgpu.ids | grep "2^16\|2^20" >> $O/r06u_ab.txt
    MZK_ROOT_MAILBOX=$v python tools/timing/fri_round_cost.py 2>&1 | grep -v amdgpu.ids >> $O/r06u_ab.txt
  done
done
cat $O/r06u_ab.txt
