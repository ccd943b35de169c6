#!/bin/bash
# round 3, job a: the launcher (python bench.py --gpus N by itself), peer access + multi-stream exchange, Merkle handle contexts,
# SRS table checksum -- on the one-GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_srs_io.py tests/test_gpu_merkle.py tests/test_gpu_cpp_mirror.py tests/test_gpu_fri_protocol.py tests/test_gpu_dev_api.py -m gpu -x -q ) > $O/r03a_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03a_pytest.log
# 1. a bare `python bench.py --gpus 2` on a one-GPU box must fail loudly with a JSON line
python bench.py --gpus 2 > $O/r03a_gpus2_on_1gpu.json 2> $O/r03a_gpus2_on_1gpu.err; echo "bare --gpus 2 rc=$?" >> $O/r03a_pytest.log
# 2. the shared-GPU rehearsal through the NEW launcher (no torch.distributed.run on the command line)
export MZK_BENCH_SHARED_GPU_TEST=1 MZK_BENCH_WATCHDOG_S=500
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --log2n 18 --extra-sizes= --e2e-log2n 18 --strong-log2n 20 --strong-ntt-log2n 20 > $O/r03a_rehearsal_n2.json 2> $O/r03a_rehearsal_n2.err
echo "rehearsal n2 rc=$?" >> $O/r03a_pytest.log
tail -5 $O/r03a_pytest.log; cat $O/r03a_gpus2_on_1gpu.json; head -c 700 $O/r03a_rehearsal_n2.json; echo; tail -3 $O/r03a_rehearsal_n2.err
