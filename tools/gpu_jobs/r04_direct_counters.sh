cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmc -- python3 $R/tools/timing/many_commit.py 10:256:1:10 > /tmp/o.txt 2>&1
tail -1 /tmp/o.txt | cut -c1-200
python3 $R/tools/timing/pmc_sq_summary.py /tmp/pmc k_direct k_seg_accumulate
