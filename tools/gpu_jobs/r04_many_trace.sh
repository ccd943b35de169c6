cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for spec in 10:256 10:256:10; do
  rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 tools/timing/many_commit.py $spec > /tmp/o.txt 2>&1
  f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
  echo "== $spec"; tail -1 /tmp/o.txt
  python3 tools/timing/trace_summary.py $f --tail 12
done
