#!/bin/bash
# round 2, job t: rehearsal of bench.py's N>1 code path at HEAD on the one-GPU box (ranks share cuda:0, exchange over gloo):
# not a measurement -- it shows the world-size-2 and -4 legs run to the JSON line with the in-flight contexts alive.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export MZK_BENCH_SHARED_GPU_TEST=1 MZK_BENCH_WATCHDOG_S=500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > $O/r02t_rehearsal_n2.json 2> $O/r02t_rehearsal_n2.err
echo "n2 rc=$?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 3 --warmup 1 --log2n 18 --extra-sizes= --e2e-log2n 18 --strong-log2n 20 > $O/r02t_rehearsal_n4.json 2> $O/r02t_rehearsal_n4.err
echo "n4 rc=$?"
tail -c 1500 $O/r02t_rehearsal_n2.json; echo; tail -3 $O/r02t_rehearsal_n2.err; tail -c 800 $O/r02t_rehearsal_n4.json; echo; tail -3 $O/r02t_rehearsal_n4.err
