#!/bin/bash
# round 5, job c: NTT-only check (tests that touch the transforms), timings, kernel durations of both fields at 2^20
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05c}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_coset_divide.py tests/test_gpu_fri_protocol.py tests/test_gpu_poly.py tests/test_gpu_field_asm.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/time_ntt.py 20,22,24 > $O/${T}_time_ntt.txt 2>&1
python tools/timing/time_lde.py > $O/${T}_time_lde.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_ntt${f}_trace -- python3 $R/tools/timing/ntt_only.py $f 20 > $O/${T}_ntt${f}_trace.log 2>&1
find $O/${T}_ntt${f}_trace -name "*kernel_stats.csv" -exec cp {} $O/${T}_ntt${f}_kernel_stats.csv \;
done
cd $R
find $O -name "*.csv" -size +4M -delete
tail -4 $O/${T}_pytest.log; grep -v amdgpu $O/${T}_time_ntt.txt; grep -v amdgpu $O/${T}_time_lde.txt | grep "2^18 -> 2^20"; python3 - <<PY
import csv
for f in (0,1):
    for r in csv.DictReader(open("$O/${T}_ntt%d_kernel_stats.csv" % f)):
        if "k_ntt" in r["Name"]: print(f, r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY
