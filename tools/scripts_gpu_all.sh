#!/bin/bash
# GPU pass: parity tests, then rocprofv3 kernel trace of a short bench. Stops if a step is killed.
mkdir -p gpurun_out/prof
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench.log 2>&1
rc=$?
echo "prof rc=$rc"
cat $GRAFT_REPO_ROOT/gpurun_out/prof/r01_kernel_stats.csv | cut -c1-200
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/prof/bench.log | cut -c1-330
exit $rc
