#!/bin/bash
# round 6, job m: the coarse scatter with ONE digit walk per chunk (fixed LDS slots per bin): parity, then A/B against the two-walk staged kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_dev_api.py tests/test_gpu_fuzz_slice.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r06m_pytest.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
rm -f $O/r06m_ab.txt
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== MZK_COARSE_SLOTS=$v (rep $rep)" >> $O/r06m_ab.txt
    MZK_COARSE_SLOTS=$v python tools/timing/commit_only.py 20 60 2>&1 | grep -v amdgpu.ids >> $O/r06m_ab.txt
    MZK_COARSE_SLOTS=$v python tools/timing/window_sweep.py 16,18,19,20 1 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $O/r06m_ab.txt
  done
done
for v in 0 1; do
  echo "== skewed scalars, MZK_COARSE_SLOTS=$v" >> $O/r06m_ab.txt
  MZK_COARSE_SLOTS=$v python tools/timing/skew_msm.py "uniform,bits,bytes,16-bit scalars,64-bit,248-bit,all ones,all-equal" 2>&1 | grep -v amdgpu.ids | grep merged >> $O/r06m_ab.txt
done
cat $O/r06m_ab.txt
