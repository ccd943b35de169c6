#!/bin/bash
# round 4: window widths 16 / 17 / 20 / 22 from 2^22 to 2^24 (the table in mzk_common.h had no 2^24 entry for c = 20, c = 22
# was never tried), after the scan's re-summing was bounded; generic 2^24 as the scan's A/B record
O=gpurun_out; mkdir -p $O
python tools/timing/window_sweep.py 22,23,24 16,17,20,22 2>&1 | grep -v amdgpu.ids | tee $O/r04_window_sweep.txt
python tools/timing/generic_phases.py 24 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $O/r04_window_sweep.txt
