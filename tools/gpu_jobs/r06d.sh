#!/bin/bash
# round 6, job d: generic layout with 19-bit windows from 2^23 pairs (parity at full sizes), the window sweep the verdict asked for,
# the scan-free sort with two-barrier block scans, a default bench run of the rewritten bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_many.py tests/test_gpu_full_size.py tests/test_gpu_dev_api.py tests/test_gpu_fuzz_slice.py tests/test_gpu_max_sizes.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r06d_pytest.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
rm -f $O/r06d_sweep.txt
for lg in 20 22 23 24 26; do
  for c in 16 17 19 20; do
    echo "== MZK_GLV_C=$c" >> $O/r06d_sweep.txt
    MZK_GLV_C=$c timeout 600 python tools/timing/generic_phases.py $lg 2>&1 | grep -v amdgpu.ids >> $O/r06d_sweep.txt
  done
done
echo "== MZK_GLV_C=22 (one-pass sort: 6 x 2^21 buckets exceed the two-level sort's bins)" >> $O/r06d_sweep.txt
MZK_GLV_C=22 timeout 600 python tools/timing/generic_phases.py 24 2>&1 | grep -v amdgpu.ids >> $O/r06d_sweep.txt
cat $O/r06d_sweep.txt
rm -f $O/r06d_sort_ab.txt
for rep in 1 2; do
for v in 0 3; do
  echo "== phases MZK_SORT_SCAN_FREE=$v (rep $rep)" >> $O/r06d_sort_ab.txt
  MZK_SORT_SCAN_FREE=$v python tools/timing/window_sweep.py 16,20,24 1 2>&1 | grep -v amdgpu.ids | cut -c1-220 >> $O/r06d_sort_ab.txt
  MZK_SORT_SCAN_FREE=$v python tools/timing/generic_phases.py 16 20 2>&1 | grep -v amdgpu.ids >> $O/r06d_sort_ab.txt
done
done
cat $O/r06d_sort_ab.txt
unset MZK_HIP_LIB
timeout 900 python bench.py --detail-file $O/r06d_bench_detail.json > $O/r06d_bench.json 2> $O/r06d_bench.err; echo "bench rc=$?"
wc -c $O/r06d_bench.json
