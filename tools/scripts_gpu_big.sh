#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "2048 or 4096 or 8192 or 16384 or sizes or detrend or stress or adcdac" > gpurun_out/pytest_big.log 2>&1
rc=$?
tail -2 gpurun_out/pytest_big.log
if [ $rc -ne 0 ]; then exit $rc; fi
for args in "--n 2048 --steps 60" "--n 4096 --steps 60" "--n 8192 --steps 40" "--n 16384 --steps 30" "--n 4096 --steps 40 --detrend mean"; do
  timeout -k 10 300 python bench.py $args --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg.log 2>&1
  rc=$?
  echo "== $args rc=$rc"
  grep '^{' gpurun_out/bench_cfg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'kernel avg ms',round(d['roofline']['avg_launch_ms'],4))" || tail -3 gpurun_out/bench_cfg.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
