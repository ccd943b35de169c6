#!/bin/bash
# round 6, job g: sharded open quotient (GPU parity), then the forced one-rank nccl record and the world-2/4/8 rehearsals of the rewritten bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_e2e_kzg.py tests/test_gpu_rccl_world1.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r06g_pytest.txt
bash tools/gpu_jobs/r06_rehearsal.sh r06g
