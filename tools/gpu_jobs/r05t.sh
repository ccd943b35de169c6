#!/bin/bash
# round 5, job t: write-through (sc1) stores in the transform passes against plain stores, same box, interleaved; which passes, which sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r05t}
mkdir -p $O
cd $R
rm -f $O/${T}_ab.txt
for rep in 1 2; do
for mask in 0 1 2 3; do
  echo "== write-through mask $mask (1: strided passes, 2: last pass), 2^16..2^30 bytes (rep $rep)" >> $O/${T}_ab.txt
  MZK_NTT_WT_LO=16 MZK_NTT_WT_HI=30 MZK_NTT_WT_MASK=$mask MZK_HIP_LIB=$R/myzkp_amd/libmzk_hip_tuning.so python tools/timing/time_ntt.py 16,17,18,19,20,21,22,23,24 2>&1 | grep -v amdgpu >> $O/${T}_ab.txt
done
done
python3 - <<PY
import re, collections
t = collections.defaultdict(lambda: collections.defaultdict(list)); mask = None
for line in open("$O/${T}_ab.txt"):
    m = re.match(r"== write-through mask (\d)", line)
    if m: mask = int(m.group(1)); continue
    m = re.match(r"(\w+) 2\^(\d+): ([\d.]+) ms", line)
    if m: t[(m.group(1), int(m.group(2)))][mask].append(float(m.group(3)))
print("min of reps, ms            mask0    mask1    mask2    mask3")
for k in sorted(t): print("%-5s 2^%-2d            " % k + "  ".join("%7.4f" % min(t[k][m]) for m in range(4)))
PY
