#!/bin/bash
# kernel time vs samples per step (fixed cost of a launch vs per-pair cost)
mkdir -p gpurun_out
for lb in 22 23 24 25 26 27 28; do
  timeout -k 10 300 python bench.py --log2-batch $lb --steps 40 --warmup 3 --no-cpu-baseline > gpurun_out/bench_sz.log 2>&1
  rc=$?
  grep '^{' gpurun_out/bench_sz.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('log2',$lb,'MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'kernel avg ms',round(d['roofline']['avg_launch_ms'],4))" || tail -3 gpurun_out/bench_sz.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
