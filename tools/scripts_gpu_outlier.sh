#!/bin/bash
# repeat a short bench and show the slowest runs with the debug timing of their read-out
mkdir -p gpurun_out
: > gpurun_out/outlier.log
for i in $(seq 1 ${REPS:-20}); do
  PSD_BENCH_DEBUG=1 timeout -k 10 120 python bench.py "$@" --no-cpu-baseline > gpurun_out/o.log 2>&1
  v=$(grep '^{' gpurun_out/o.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
  echo "$v $(grep '^\[debug\]' gpurun_out/o.log)" >> gpurun_out/outlier.log
done
sort -n gpurun_out/outlier.log | head -4
echo ...
sort -n gpurun_out/outlier.log | tail -2
