#!/bin/bash
# round 5, job f: transform tests, same-box timing against the round-4 library, SQ counters of the M128 and Fr passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05f}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_coset_divide.py tests/test_gpu_fri_protocol.py tests/test_gpu_poly.py tests/test_gpu_field_asm.py tests/test_gpu_sharded_ntt.py tests/test_gpu_full_size.py -m gpu -x -q -k "not msm" ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
for rep in 1 2; do
for lib in libmzk_hip_r04.so libmzk_hip.so; do
  [ -f myzkp_amd/$lib ] || continue
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_ntt.py 10,14,18,20,22,24 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
done
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"
for f in 0 1; do
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_ntt${f}_SQ -- python3 $R/tools/timing/ntt_only.py $f 20 > $O/${T}_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_ntt${f}_trace -- python3 $R/tools/timing/ntt_only.py $f 20 > $O/${T}_trace.log 2>&1
find $O/${T}_ntt${f}_trace -name "*kernel_stats.csv" -exec cp {} $O/${T}_ntt${f}_kernel_stats.csv \;
done
{ echo "== NTT Fr 2^20"; python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_ntt0_SQ k_ntt; echo "== NTT M128 2^20"; python3 $R/tools/timing/pmc_sq_summary.py $O/${T}_ntt1_SQ k_ntt; } > $O/${T}_sq_counters.txt 2>&1
cd $R
find $O -name "*.csv" -size +4M -delete
tail -4 $O/${T}_pytest.log; cat $O/${T}_ab.txt; grep "k_ntt\|SQ_INSTS_VALU\|SQ_WAVE_CYCLES\|SQ_WAIT" $O/${T}_sq_counters.txt | cut -c1-110; python3 - <<PY
import csv
for f in (0,1):
    for r in csv.DictReader(open("$O/${T}_ntt%d_kernel_stats.csv" % f)):
        if "k_ntt" in r["Name"]: print(f, r["Name"][10:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY
