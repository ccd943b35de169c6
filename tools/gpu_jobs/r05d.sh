#!/bin/bash
# round 5, job d: same-box A/B of the transform kernels: round-4 library / this tree / this tree with early inter-pass twiddles
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05d}
mkdir -p $O
cd $R
for rep in 1 2; do
for lib in libmzk_hip_r04.so libmzk_hip.so libmzk_hip_earlytw.so; do
  [ -f myzkp_amd/$lib ] || continue
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_ntt.py 20,24 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
done
cd /tmp && export TMPDIR=/tmp
for lib in libmzk_hip_r04.so libmzk_hip.so; do
for f in 0 1; do
export MZK_HIP_LIB=$R/myzkp_amd/$lib
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_${lib}_ntt${f} -- python3 $R/tools/timing/ntt_only.py $f 20 > $O/${T}_trace.log 2>&1
find $O/${T}_${lib}_ntt${f} -name "*kernel_stats.csv" -exec cp {} $O/${T}_${lib}_ntt${f}_kernel_stats.csv \;
done
done
unset MZK_HIP_LIB
cd $R
find $O -name "*.csv" -size +4M -delete
cat $O/${T}_ab.txt; python3 - <<PY
import csv, glob
for fn in sorted(glob.glob("$O/${T}_*_kernel_stats.csv")):
    for r in csv.DictReader(open(fn)):
        if "k_ntt" in r["Name"]: print(fn.split("/")[-1][5:35], r["Name"][10:58], r["Calls"], r["AverageNs"], r["MinNs"])
PY
