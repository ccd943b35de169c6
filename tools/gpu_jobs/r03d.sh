#!/bin/bash
# round 3, job d: row-cooperative tails (mzk_row.h): self-test, MSM parity subset, A/B latency against the quad tails
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_row_ec.py tests/test_gpu_msm.py tests/test_gpu_dev_api.py tests/test_gpu_e2e_kzg.py tests/test_gpu_multi.py -m gpu -x -q ) > $O/r03d_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03d_pytest.log
for f in 0 1; do
  echo "== MZK_ROW_TAILS=$f" >> $O/r03d_latency.txt
  MZK_ROW_TAILS=$f timeout 600 python tools/timing/small_latency.py 4,10,12,13,14,16,18,20 >> $O/r03d_latency.txt 2>&1
  MZK_ROW_TAILS=$f timeout 600 python tools/timing/commit_only.py 20 40 >> $O/r03d_latency.txt 2>&1
done
tail -12 $O/r03d_pytest.log; cat $O/r03d_latency.txt
