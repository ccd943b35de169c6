#!/bin/bash
# A/B two library builds in one GPU session: $1 = variant suffix (myzkp_amd/libmzk_hip_<suffix>.so)
cd $GRAFT_REPO_ROOT
cp myzkp_amd/libmzk_hip.so /tmp/orig.so
for V in orig $1 orig $1; do
  if [ $V = orig ]; then cp /tmp/orig.so myzkp_amd/libmzk_hip.so; else cp myzkp_amd/libmzk_hip_$V.so myzkp_amd/libmzk_hip.so; fi
  python bench.py --skip-cpu --extra-sizes "" --e2e-log2n 0 --strong-log2n 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$V', round(d['ms_per_step'],4), {k:round(v['avg_ms'],3) for k,v in d['phases'].items()}, 'generic', round(d['msm_generic']['ms_per_step'],3), round(d['msm_generic']['phases']['msm_bucket_accumulate']['avg_ms'],3))"
done
cp /tmp/orig.so myzkp_amd/libmzk_hip.so
