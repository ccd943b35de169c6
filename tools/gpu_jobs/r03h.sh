#!/bin/bash
# round 3, job h: pinned staging ring for the host-buffer entry points (A/B against the runtime's pageable path), then the full
# GPU suite and the default bench line on this tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for f in 0 1; do
  echo "== MZK_STAGE=$f (0 = hipMemcpyAsync from pageable memory, 1 = pinned ring + 4 copy threads)" >> $O/r03h_pcie.txt
  MZK_STAGE=$f timeout 600 python tools/timing/pcie_incl.py >> $O/r03h_pcie.txt 2>&1
done
grep -v amdgpu.ids $O/r03h_pcie.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/r03h_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03h_pytest.log
tail -12 $O/r03h_pytest.log
timeout 1200 python bench.py > $O/r03h_bench.json 2> $O/r03h_bench.err
echo "bench rc=$?"; head -c 400 $O/r03h_bench.json; echo
