#!/bin/bash
# A/B two library builds in one GPU session: myzkp_amd/libmzk_hip_old.so vs the current one
cd $GRAFT_REPO_ROOT
cp myzkp_amd/libmzk_hip.so /tmp/new.so
for v in old new old new; do
  cp myzkp_amd/libmzk_hip_old.so myzkp_amd/libmzk_hip.so; [ $v = new ] && cp /tmp/new.so myzkp_amd/libmzk_hip.so
  python bench.py --steps 10 --warmup 2 --skip-cpu --extra-sizes "" --e2e-log2n 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ph=d['phases']
print('$v', 'SRS', round(d['ms_per_step'],3), {k[4:]:round(v['avg_ms'],3) for k,v in ph.items()}, '| generic', round(d['msm_generic']['ms_per_step'],3), '| ntt', round(d['ntt']['ms_per_step'],4), 'm128', round(d['ntt_m128']['ms_per_step'],4))"
done
cp /tmp/new.so myzkp_amd/libmzk_hip.so
