#!/bin/bash
# round 2, job v: the sharded-transform leg of bench.py: N = 1 (plain), and rehearsals at world 2 / 4 / 8 with the ranks sharing
# cuda:0 and the exchanges over gloo (not measurements: they show the leg runs and every part equals the single-GPU transform)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
ARGS="--steps 3 --warmup 1 --log2n 16 --extra-sizes= --e2e-log2n 0 --strong-log2n 0 --skip-cpu --no-two-in-flight"
timeout 600 python bench.py --gpus 1 $ARGS > $O/r02v_n1.json 2> $O/r02v_n1.err; echo "n1 rc=$?"
export MZK_BENCH_SHARED_GPU_TEST=1 MZK_BENCH_WATCHDOG_S=400
for W in 2 4 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29520 + W)) bench.py --gpus $W $ARGS > $O/r02v_n$W.json 2> $O/r02v_n$W.err
  echo "n$W rc=$?"
done
for W in 1 2 4 8; do python - <<PY
import json
try:
    d = json.loads(open("$O/r02v_n$W.json").read().strip().splitlines()[-1])
    print($W, json.dumps(d.get("strong_scaling_ntt")))
except Exception as e:
    print($W, "no json", e); print(open("$O/r02v_n$W.err").read()[-1500:])
PY
done
