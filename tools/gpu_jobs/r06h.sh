#!/bin/bash
# round 6, job h: quads per bucket in the grid-batched segment combine (parity + A/B), then the rehearsals again (the clock leg of
# job g ran its collective on rank 0 only and hung every N > 1 run until its timeout)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_many.py tests/test_gpu_msm.py tests/test_gpu_fuzz_slice.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r06h_pytest.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
rm -f $O/r06h_qpb.txt
for rep in 1 2; do
  for q in 1 2 4 0; do
    echo "== MZK_COMBINE_QPB=$q (0 = the library's choice) rep $rep" >> $O/r06h_qpb.txt
    MZK_COMBINE_QPB=$q python tools/timing/many_commit.py 10:256,12:64,14:16,13:32,10:256:1:0:248,11:128 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O/r06h_qpb.txt
  done
done
cat $O/r06h_qpb.txt
unset MZK_HIP_LIB
bash tools/gpu_jobs/r06_rehearsal.sh r06h
