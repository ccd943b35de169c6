#!/bin/bash
# round 6, job r: the open stage of the end-to-end leg taken apart
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python tools/timing/e2e_open_split.py 22 2>&1 | grep -v amdgpu.ids | tee $O/r06r_open_split.txt
