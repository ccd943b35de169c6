#!/bin/bash
# round 6, job p: after the subgroup-prefix interpolation -- two fuzz seeds (40 % of the polynomial cases now draw such domains) and the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tools/gpu_jobs/r06_fuzz.sh 6201 6202 > /dev/null 2>&1
cp $O/r06_differential_fuzz.txt $O/r06p_fuzz.txt
timeout 1200 python bench.py --detail-file $O/r06p_bench_default_detail.json > $O/r06p_bench_default.json 2> $O/r06p_bench.err
echo "bench rc=$?"
sed "s/: .*; /: ... /" $O/r06p_fuzz.txt
head -c 1500 $O/r06p_bench_default.json
