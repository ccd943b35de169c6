#!/bin/bash
# round 6, job o: interpolation over a prefix of a power-of-two subgroup (the STARK trace domain) as one inverse transform: parity of the polynomial
# routines and their callers, then the stage times with the path on (shipped) and off (tuning build, MZK_INTERP_PREFIX=0: the subproduct tree)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_poly.py tests/test_gpu_fri_protocol.py tests/test_gpu_next_rows.py tests/test_gpu_dev_api.py tests/test_gpu_cpp_mirror.py -m gpu -x -q 2>&1 | tail -15 | tee $O/r06o_pytest.txt
rm -f $O/r06o_ab.txt
for rep in 1 2; do
  echo "== shipped library (rep $rep)" >> $O/r06o_ab.txt
  python tools/timing/interp_dev_time.py 2>&1 | grep -v amdgpu.ids >> $O/r06o_ab.txt
  echo "== tuning build, MZK_INTERP_PREFIX=0 (rep $rep)" >> $O/r06o_ab.txt
  MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so MZK_INTERP_PREFIX=0 python tools/timing/interp_dev_time.py 2>&1 | grep -v amdgpu.ids >> $O/r06o_ab.txt
done
echo "== stark_commit_pipeline.py (shipped library)" >> $O/r06o_ab.txt
python tools/timing/stark_commit_pipeline.py 14 16 2>&1 | grep -v amdgpu.ids | head -40 >> $O/r06o_ab.txt
cat $O/r06o_ab.txt
