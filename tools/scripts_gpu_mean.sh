#!/bin/bash
# Mean detrend: parity subset, then bench lines at N = 256 / 512 / 1024 (and the big kernels with "big")
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "detrend or stress or golden or large_dc or settings" > gpurun_out/pytest_mean.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_mean.log
if [ $rc -ne 0 ]; then exit $rc; fi
sizes="256 512 1024"
if [ "$1" = "big" ]; then sizes="$sizes 2048 4096 8192 16384"; fi
for n in $sizes; do
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --n $n --detrend mean --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/bench_mean_$n.log 2>&1 || exit 1
    grep '^{' gpurun_out/bench_mean_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N', $n, 'mean: MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4))"
  done
done
