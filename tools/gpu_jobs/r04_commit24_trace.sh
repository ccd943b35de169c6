cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 $R/tools/timing/commit_only.py 24 6 > /tmp/o.txt 2>&1
tail -1 /tmp/o.txt
python3 $R/tools/timing/trace_summary.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) --tail 27
