#!/bin/bash
cd $GRAFT_REPO_ROOT
cp myzkp_amd/libmzk_hip.so /tmp/orig.so
for v in orig inline; do
  if [ $v = inline ]; then cp myzkp_amd/libmzk_hip_inline.so myzkp_amd/libmzk_hip.so; fi
  echo "== $v"
  python bench.py --steps 10 --warmup 2 --skip-cpu --extra-sizes "" --e2e-log2n 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('SRS', round(d['ms_per_step'],3), 'combine', round(d['phases']['msm_window_combine']['avg_ms'],3), '| generic', round(d['msm_generic']['ms_per_step'],3), 'combine', round(d['msm_generic']['phases']['msm_window_combine']['avg_ms'],3))"
done
cp /tmp/orig.so myzkp_amd/libmzk_hip.so
