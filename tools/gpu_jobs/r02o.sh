#!/bin/bash
# round 2, job o: HBM-side traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) of the current MSM and NTT kernels,
# on the small single-purpose drivers (the whole bench under --pmc takes a quarter of an hour)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r02o_msm_$c -- python3 $R/tools/timing/acc_sweep.py --child 20 > $O/r02o_msm_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r02o_ntt_$c -- python3 $R/tools/timing/ntt_only.py 0 20 > $O/r02o_ntt_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r02o_nttm_$c -- python3 $R/tools/timing/ntt_only.py 1 20 > $O/r02o_nttm_$c.log 2>&1
done
{ echo "== KZG commit 2^20 (acc_sweep.py --child 20)"; python3 $R/tools/timing/pmc_summary.py $O/r02o_msm_FETCH_SIZE $O/r02o_msm_WRITE_SIZE;
  echo "== Fr NTT 2^20 (ntt_only.py 0 20)"; python3 $R/tools/timing/pmc_summary.py $O/r02o_ntt_FETCH_SIZE $O/r02o_ntt_WRITE_SIZE;
  echo "== M128 NTT 2^20 (ntt_only.py 1 20)"; python3 $R/tools/timing/pmc_summary.py $O/r02o_nttm_FETCH_SIZE $O/r02o_nttm_WRITE_SIZE; } > $O/r02o_hbm_traffic_pmc.txt 2>&1
find $O -path "*r02o_*" -name "*.csv" -size +3M -delete
cat $O/r02o_hbm_traffic_pmc.txt
