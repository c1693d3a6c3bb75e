#!/bin/bash
# rocprofv3 kernel trace of the default bench workload (short)
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench.log 2>&1
rc=$?
echo "rc=$rc"
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -30
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof/bench.log | cut -c1-400
exit $rc
