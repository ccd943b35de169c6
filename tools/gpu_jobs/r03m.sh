#!/bin/bash
# round 3, job m: k_small_accumulate with a lane per entry: MSM parity subset + latency table + kernel trace of small commits
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_row_ec.py tests/test_gpu_msm.py tests/test_gpu_dev_api.py tests/test_gpu_multi.py tests/test_gpu_e2e_kzg.py tests/test_gpu_next_rows.py tests/test_gpu_srs_io.py -m gpu -x -q ) > $O/r03m_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03m_pytest.log
tail -6 $O/r03m_pytest.log
timeout 600 python tools/timing/small_latency.py 4,8,10,11,12,13,14,16 > $O/r03m.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03m_small -- python3 $R/tools/timing/small_trace.py > $O/r03m_small.log 2>&1
python3 $R/tools/timing/small_trace_summary.py $O/r03m_small >> $O/r03m.txt 2>&1
find $O -path "*r03m_*" -name "*.csv" -size +3M -delete
grep -v amdgpu.ids $O/r03m.txt | head -40
