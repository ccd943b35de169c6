#!/bin/bash
# round 6, job c: the sort without global scans (bin runs and bucket runs claimed with atomics): parity, same-box A/B, then the
# rewritten bench.py at a small size and through the one-rank nccl group
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_many.py tests/test_gpu_full_size.py tests/test_gpu_dev_api.py tests/test_gpu_fuzz_slice.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r06c_pytest.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
rm -f $O/r06c_ab.txt
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    echo "== MZK_SORT_SCAN_FREE=$v (rep $rep)" >> $O/r06c_ab.txt
    MZK_SORT_SCAN_FREE=$v python tools/timing/commit_only.py 20 40 2>&1 | grep -v amdgpu.ids >> $O/r06c_ab.txt
  done
done
for v in 0 3; do
  echo "== phases MZK_SORT_SCAN_FREE=$v" >> $O/r06c_ab.txt
  MZK_SORT_SCAN_FREE=$v python tools/timing/window_sweep.py 16,18,20,22,24 1 2>&1 | grep -v amdgpu.ids | cut -c1-220 >> $O/r06c_ab.txt
  MZK_SORT_SCAN_FREE=$v python tools/timing/generic_phases.py 16 20 24 2>&1 | grep -v amdgpu.ids >> $O/r06c_ab.txt
  MZK_SORT_SCAN_FREE=$v python tools/timing/skew_msm.py 2>&1 | grep -v amdgpu.ids | tail -12 >> $O/r06c_ab.txt
done
cat $O/r06c_ab.txt
unset MZK_HIP_LIB
MZK_BENCH_VERBOSE=1 timeout 900 python bench.py --log2n 16 --e2e-log2n 16 --strong-log2n 16 --strong-ntt-log2n 16 --extra-sizes 18 --steps 3 --detail-file $O/r06c_bench_detail_small.json > $O/r06c_bench_small.json 2> $O/r06c_bench_small.err; echo "bench rc=$?"
tail -c 3000 $O/r06c_bench_small.err | grep -v amdgpu.ids | tail -15
wc -c $O/r06c_bench_small.json
timeout 1500 python -m pytest tests/test_gpu_rccl_world1.py -m gpu -x -q 2>&1 | tail -15 | tee $O/r06c_pytest_rccl.txt
