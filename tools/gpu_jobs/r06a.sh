#!/bin/bash
# round 6, job a: the RCCL branch at world 1 (first execution anywhere), then this box's baseline of the r05 tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python tests/rccl_world1_child.py > $O/r06a_rccl_child.txt 2>&1; echo "rc=$?" >> $O/r06a_rccl_child.txt
tail -5 $O/r06a_rccl_child.txt
for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/pp_dpm_mclk /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input; do echo "== $f"; cat $f; done > $O/r06a_sysfs.txt 2>&1
cat $O/r06a_sysfs.txt
for rep in 1 2; do
  python tools/timing/commit_only.py 20 40 2>&1 | grep -v amdgpu.ids
  python tools/timing/generic_phases.py 20 24 2>&1 | grep -v amdgpu.ids
  python tools/timing/commit_only.py 24 10 2>&1 | grep -v amdgpu.ids
done | tee $O/r06a_baseline.txt
