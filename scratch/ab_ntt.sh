#!/bin/bash
# A/B different library builds in one GPU session
cd $GRAFT_REPO_ROOT
cp myzkp_amd/libmzk_hip.so /tmp/orig.so
for V in orig lv9 lv10 orig; do
  if [ $V = orig ]; then cp /tmp/orig.so myzkp_amd/libmzk_hip.so; else cp myzkp_amd/libmzk_hip_$V.so myzkp_amd/libmzk_hip.so; fi
  echo "== variant $V"
  python scratch/time_ntt.py 2>&1 | grep -v amdgpu.ids | grep "2^20"
done
cp /tmp/orig.so myzkp_amd/libmzk_hip.so
