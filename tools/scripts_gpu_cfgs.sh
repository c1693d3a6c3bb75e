#!/bin/bash
mkdir -p gpurun_out
for args in "--channels-per-gpu 1 --log2-batch 23 --steps 100" "--channels-per-gpu 2 --log2-batch 25 --steps 60" "--channels-per-gpu 4 --log2-batch 24 --steps 60" "--channels-per-gpu 8 --log2-batch 23 --steps 60" "--channels-per-gpu 8 --log2-batch 24 --steps 40"; do
  timeout -k 10 300 python bench.py $args --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg.log 2>&1
  rc=$?
  echo "== $args rc=$rc"
  grep '^{' gpurun_out/bench_cfg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'host',round(d['host_enqueue_ms_per_step'],4),'kernel avg ms',round(d['roofline']['avg_launch_ms'],4),'launches',d['roofline']['launches'], d['config']['stages'],'stages')" || tail -3 gpurun_out/bench_cfg.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
