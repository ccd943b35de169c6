#!/bin/bash
# round 5, job n: 1024 coarse bins of the sort at 20-bit windows: full-size tests, same-box A/B in the tuning build (MZK_COARSE_LOG_20 = 8 | 10),
# kernel trace of the 2^24 commit
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05n}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_msm.py tests/test_gpu_dev_api.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
for rep in 1 2; do
for cl in 8 10; do
  echo "== MZK_COARSE_LOG_20=$cl (rep $rep)" >> $O/${T}_ab.txt
  MZK_COARSE_LOG_20=$cl python tools/timing/commit_only.py 22 12 2>&1 | grep -v amdgpu | tail -1 >> $O/${T}_ab.txt
  MZK_COARSE_LOG_20=$cl python tools/timing/commit_only.py 24 8 2>&1 | grep -v amdgpu | tail -1 >> $O/${T}_ab.txt
done
done
unset MZK_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 $R/tools/timing/commit_only.py 24 6 > /tmp/o.txt 2>&1
python3 $R/tools/timing/trace_summary.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) --tail 30 > $O/${T}_commit24_trace.txt 2>&1
cd $R
tail -4 $O/${T}_pytest.log; cat $O/${T}_ab.txt; cat $O/${T}_commit24_trace.txt | cut -c1-150
