#!/bin/bash
# round 3, job 4s: run-to-run spread of the headline with the default (10 steps, 2 warm-up) and with 30 steps / 10 warm-up
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
LEAN="--skip-cpu --extra-sizes= --e2e-log2n 0 --strong-log2n 0 --strong-ntt-log2n 0 --no-two-in-flight"
for rep in 1 2 3 4; do
  for cfg in "--steps 10 --warmup 2" "--steps 30 --warmup 10"; do
    python bench.py $cfg $LEAN 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', 'ms_per_step %.4f' % d['ms_per_step'], 'acc %.4f' % d['phases']['msm_bucket_accumulate_in_timed_region']['avg_ms'], 'ntt %.4f' % d['ntt']['ms_per_step'], 'generic %.4f' % d['msm_generic']['ms_per_step'])
" | tee -a $O/r04s_bench_spread.txt
  done
done
