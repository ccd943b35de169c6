#!/bin/bash
# round 5, job ag: polynomial tests after the k_chunk_combine rewrite; interpolation stage trace
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05ag}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_poly.py tests/test_gpu_coset_divide.py tests/test_gpu_fri_protocol.py tests/test_gpu_dev_api.py -m gpu -x -q ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python tools/timing/stark_commit_pipeline.py 14 16 2>&1 | grep -v amdgpu > $O/${T}_pipeline.txt
python tools/timing/stark_commit_pipeline.py 12 16 2>&1 | grep -v amdgpu >> $O/${T}_pipeline.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_interp -- python3 $R/tools/timing/stark_stage_trace.py interp 14 16 4 > $O/${T}_interp.log 2>&1
find $O/${T}_interp -name "*kernel_stats.csv" -exec cp {} $O/${T}_interp_kernel_stats.csv \;
cd $R
find $O -name "*.csv" -size +4M -delete
tail -4 $O/${T}_pytest.log; cat $O/${T}_pipeline.txt; grep "interp rep" $O/${T}_interp.log; head -8 $O/${T}_interp_kernel_stats.csv | cut -c1-60,150-230
timeout 600 python bench.py --steps 5 --warmup 1 --skip-cpu --extra-sizes= --e2e-log2n 0 --strong-log2n 0 --strong-ntt-log2n 0 > $O/${T}_bench.json 2> $O/${T}_bench.err
python3 -c "
import json
d=json.loads(open('$O/${T}_bench.json').read().strip().split('\n')[-1])
print(json.dumps(d['stark_commit_pipeline'], indent=1)[:1800])"
