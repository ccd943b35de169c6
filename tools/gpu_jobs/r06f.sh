#!/bin/bash
# round 6, job f: host-buffer entry points in pieces (upload of piece k + 1 under the kernels of piece k): parity, then the
# PCIe-inclusive times one piece vs pieces, and the verdict's two-half pipeline as a what-if on resident coefficients
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_dev_api.py tests/test_gpu_e2e_kzg.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r06f_pytest.txt
export MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so
rm -f $O/r06f_host.txt
for rep in 1 2; do
  echo "== MZK_HOST_CHUNKS=0 (one piece, rounds 2-5) rep $rep" >> $O/r06f_host.txt
  MZK_HOST_CHUNKS=0 python tools/timing/pcie_incl.py 2>&1 | grep -v amdgpu.ids >> $O/r06f_host.txt
  echo "== pieces: default (commit 25 % + 75 %, MSM four quarters) rep $rep" >> $O/r06f_host.txt
  python tools/timing/pcie_incl.py 2>&1 | grep -v amdgpu.ids >> $O/r06f_host.txt
  for plan in "32,256" "64,160,256" "128,256" "64,128,192,256"; do
    echo "== commit pieces at $plan / 256" >> $O/r06f_host.txt
    MZK_HOST_CHUNKS_COMMIT=$plan python tools/timing/pcie_incl.py commit 2>&1 | grep -v amdgpu.ids >> $O/r06f_host.txt
  done
  for plan in "128,256" "64,160,256" "32,96,176,256" "32,64,96,128,160,192,224,256"; do
    echo "== msm pieces at $plan / 256" >> $O/r06f_host.txt
    MZK_HOST_CHUNKS_MSM=$plan python tools/timing/pcie_incl.py msm 2>&1 | grep -v amdgpu.ids >> $O/r06f_host.txt
  done
done
cat $O/r06f_host.txt
rm -f $O/r06f_two_half.txt
for rep in 1 2 3; do
  for cfg in "0 0" "2 0" "2 1" "4 1"; do
    set -- $cfg
    echo "== MZK_DEV_CHUNKS=$1 MZK_DEV_SORT_STREAM=$2 (rep $rep)" >> $O/r06f_two_half.txt
    MZK_DEV_CHUNKS=$1 MZK_DEV_SORT_STREAM=$2 python tools/timing/commit_only.py 20 40 2>&1 | grep -v amdgpu.ids >> $O/r06f_two_half.txt
  done
done
cat $O/r06f_two_half.txt
