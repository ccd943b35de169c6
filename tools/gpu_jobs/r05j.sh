#!/bin/bash
# round 5, job j: the scheduled lane-pair Keccak round: Merkle / FRI tests, same-box A/B against the build before it (libmzk_hip_prekeccak.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05j}
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri_protocol.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
for rep in 1 2; do
for lib in libmzk_hip_prekeccak.so libmzk_hip.so; do
  [ -f myzkp_amd/$lib ] || continue
  echo "== $lib (rep $rep)" >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/time_merkle.py 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/$lib python tools/timing/fri_round_cost.py 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_fri -- python3 $R/tools/timing/stark_stage_trace.py fri 14 16 4 > $O/${T}_fri.log 2>&1
find $O/${T}_fri -name "*kernel_stats.csv" -exec cp {} $O/${T}_fri_kernel_stats.csv \;
cd $R
tail -3 $O/${T}_pytest.log; cat $O/${T}_ab.txt; head -6 $O/${T}_fri_kernel_stats.csv | cut -c1-150
