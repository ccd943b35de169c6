#!/bin/bash
# round 3, job l: wave-cooperative inversion (mzk_inv_wave.h): self-test, MSM parity subset, conversion cost, latency table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_row_ec.py tests/test_gpu_msm.py tests/test_gpu_dev_api.py tests/test_gpu_multi.py tests/test_gpu_e2e_kzg.py -m gpu -x -q ) > $O/r03l_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03l_pytest.log
tail -6 $O/r03l_pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03l_fold -- python3 $R/tools/timing/fold_one.py > $O/r03l_fold.log 2>&1
grep "fold of" $O/r03l_fold.log > $O/r03l.txt
find $O/r03l_fold -name "*kernel_stats.csv" -exec grep -h "fold_partials" {} \; >> $O/r03l.txt
cd $R
timeout 600 python tools/timing/small_latency.py 4,10,12,14,16,18,20 >> $O/r03l.txt 2>&1
grep -v amdgpu.ids $O/r03l.txt
