#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "" 4 5 6 7 8 ""; do
  if [ -z "$V" ]; then unset MZK_DEBUG_LGSEG; else export MZK_DEBUG_LGSEG=$V; fi; python bench.py --skip-cpu --extra-sizes "" --e2e-log2n 0 --strong-log2n 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lgseg=$V', round(d['ms_per_step'],4), {k:round(v['avg_ms'],3) for k,v in d['phases'].items()}, 'generic', round(d['msm_generic']['ms_per_step'],3), round(d['msm_generic']['phases']['msm_bucket_accumulate']['avg_ms'],3))"
done
