#!/bin/bash
# PMC passes (own runs, kernel-trace only besides --pmc) for HBM traffic of the bench kernels
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc/bench_$c.log 2>&1
  rc=$?
  echo "$c rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc/bench_$c.log; exit $rc; fi
done
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc
