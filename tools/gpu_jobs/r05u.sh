#!/bin/bash
# round 5, job u: the shipped write-through rule (template flag) against plain stores, same box, interleaved; transform tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05u}
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_coset_divide.py tests/test_gpu_fri_protocol.py tests/test_gpu_poly.py tests/test_gpu_sharded_ntt.py tests/test_gpu_full_size.py -m gpu -x -q -k "not msm" ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
rm -f $O/${T}_ab.txt
for rep in 1 2 3; do
for mask in 0 3; do
  echo "== write-through mask $mask (0: plain stores, 3: the shipped rule) (rep $rep)" >> $O/${T}_ab.txt
  MZK_NTT_WT_MASK=$mask MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so python tools/timing/time_ntt.py 16,17,18,19,20,21,22,23,24 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
done
tail -4 $O/${T}_pytest.log
python3 - <<PY
import re, collections
t = collections.defaultdict(lambda: collections.defaultdict(list)); mask = None
for line in open("$O/${T}_ab.txt"):
    m = re.match(r"== write-through mask (\d)", line)
    if m: mask = int(m.group(1)); continue
    m = re.match(r"(\w+) 2\^(\d+): ([\d.]+) ms", line)
    if m: t[(m.group(1), int(m.group(2)))][mask].append(float(m.group(3)))
print("min / median of 3 reps, ms     plain            shipped rule")
for k in sorted(t): print("%-5s 2^%-2d            " % k + "    ".join("%7.4f %7.4f" % (min(t[k][m]), sorted(t[k][m])[1]) for m in (0, 3)))
PY
