#!/bin/bash
# one line per workload shape (all on one GPU, input resident in HBM)
mkdir -p gpurun_out
for args in "--n 256" "--n 512" "--n 1024" "--n 2048" "--n 4096" "--n 8192" "--n 16384" \
            "--n 1024 --detrend midpoint" "--n 1024 --detrend span" "--n 1024 --detrend mean" "--n 4096 --detrend mean" "--n 1024 --avg 1000,100000" \
            "--n 1024 --channels-per-gpu 2 --log2-batch 25" "--n 1024 --channels-per-gpu 4 --log2-batch 24" \
            "--n 1024 --channels-per-gpu 8 --log2-batch 24" "--n 1024 --channels-per-gpu 8 --log2-batch 23"; do
  timeout -k 10 300 python bench.py $args --steps 1000 --warmup 10 --no-cpu-baseline > gpurun_out/bench_shape.log 2>&1
  rc=$?
  grep '^{' gpurun_out/bench_shape.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', '| MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'kernel avg ms',round(d['roofline']['avg_launch_ms'],4), d['config']['stages'],'stages')" || { echo "$args rc=$rc"; tail -3 gpurun_out/bench_shape.log; }
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
