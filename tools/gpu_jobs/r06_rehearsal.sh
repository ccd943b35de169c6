#!/bin/bash
# round 6: (1) the sharded legs of bench.py through a one-rank `nccl` (= RCCL) process group on this one-GPU box, (2) rehearsals of the
# N > 1 code path with all ranks on cuda:0 and the exchange over gloo (tagged REHEARSAL_NOT_A_MEASUREMENT) -- worlds 2, 4 and 8; world 8
# with configs[3]'s shard size (2^24 pairs over 8 ranks = 2^21 per rank)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
T=${1:-r06final}
timeout 900 python bench.py --gpus 1 --force-process-group --sharded-legs-only --steps 5 --warmup 2 --skip-cpu --extra-sizes= \
  --detail-file $O/${T}_bench_forced_nccl_world1_detail.json > $O/${T}_bench_forced_nccl_world1.json 2> $O/${T}_bench_forced_nccl_world1.err
echo "forced nccl rc=$?"; tail -c 400 $O/${T}_bench_forced_nccl_world1.json
export MZK_BENCH_SHARED_GPU_TEST=1 MZK_BENCH_WATCHDOG_S=700
for w in 2 4 8; do
  s=20; [ $w = 8 ] && s=24
  timeout 600 python bench.py --gpus $w --steps 3 --warmup 1 --log2n 18 --extra-sizes= --e2e-log2n 18 --strong-log2n $s --strong-ntt-log2n 20 --skip-cpu \
    --detail-file $O/${T}_rehearsal_world${w}_shared_gpu_detail.json > $O/${T}_rehearsal_world${w}_shared_gpu.json 2> $O/${T}_rehearsal_world${w}.err
  echo "rehearsal world $w rc=$?"
  tail -c 300 $O/${T}_rehearsal_world${w}_shared_gpu.json; echo
done
