#!/bin/bash
# round 6, job j: e2e tests after the world-1 fast path of the sharded quotient, the default bench line again (its traffic records are this
# tree's now), the forced one-rank nccl run, then four fuzz seeds on the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_e2e_kzg.py tests/test_gpu_rccl_world1.py -m gpu -x -q 2>&1 | tail -4 | tee $O/r06j_pytest.txt
timeout 1200 python bench.py --detail-file $O/r06j_bench_default_detail.json > $O/r06j_bench_default.json 2> $O/r06j_bench.err; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --force-process-group --sharded-legs-only --steps 5 --warmup 2 --skip-cpu --extra-sizes= \
  --detail-file $O/r06j_bench_forced_nccl_world1_detail.json > $O/r06j_bench_forced_nccl_world1.json 2> $O/r06j_forced.err; echo "forced rc=$?"
bash tools/gpu_jobs/r06_fuzz.sh 6101 6102 6103 6104
