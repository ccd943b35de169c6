#!/bin/bash
# round 5 (same as r04_rehearsal.sh, this tree): the N > 1 bench path rehearsed through bench.py's own launcher on ONE GPU (all ranks share cuda:0, exchanges over gloo
# through host memory: tagged REHEARSAL_NOT_A_MEASUREMENT) -- worlds 2, 4 and 8; world 8 with configs[3]'s shard size (2^24 pairs
# over 8 ranks = 2^21 per rank)
O=gpurun_out; mkdir -p $O
export MZK_BENCH_SHARED_GPU_TEST=1 MZK_BENCH_WATCHDOG_S=700
for w in 2 4 8; do
  s=20; [ $w = 8 ] && s=24
  timeout 1200 python bench.py --gpus $w --steps 3 --warmup 1 --log2n 18 --extra-sizes= --e2e-log2n 18 --strong-log2n $s --strong-ntt-log2n 20 --skip-cpu \
    > $O/r05final_rehearsal_world${w}_shared_gpu.json 2> $O/r05final_rehearsal_world${w}.err
  echo "rehearsal world $w rc=$?"
  tail -c 600 $O/r05final_rehearsal_world${w}_shared_gpu.json | head -c 600; echo
done
