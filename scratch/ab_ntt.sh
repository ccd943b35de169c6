#!/bin/bash
# A/B different library builds in one GPU session
cd $GRAFT_REPO_ROOT
cp myzkp_amd/libmzk_hip.so /tmp/orig.so
for T in 256 512 1024; do
  cp myzkp_amd/libmzk_hip_t$T.so myzkp_amd/libmzk_hip.so
  echo "== threads=$T"
  MZK_NTT_RB=0 python scratch/time_ntt.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/orig.so myzkp_amd/libmzk_hip.so
