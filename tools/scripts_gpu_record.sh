#!/bin/bash
# the record of a build: parity tests, default bench line, rocprofv3 kernel stats of the same command,
# FETCH_SIZE / WRITE_SIZE passes.  Everything lands in gpurun_out/record/.
out=$GRAFT_REPO_ROOT/gpurun_out/record
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -2 $out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > $out/bench_default.log 2>&1
rc=$?
echo "bench rc=$rc"; grep '^{' $out/bench_default.log | cut -c1-600
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/bench_stats.log 2>&1
rc=$?
echo "stats rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for c in FETCH_SIZE WRITE_SIZE; do
  # one span per launch here, so that a launch's traffic compares with 4 B x 2^26 samples
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --coalesce 1 --no-cpu-baseline > $out/bench_$c.log 2>&1
  rc=$?
  echo "$c rc=$rc"
  if [ $rc -ne 0 ]; then exit $rc; fi
done
cut -c1-160 $out/stats_kernel_stats.csv
