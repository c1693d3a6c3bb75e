#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 50 --warmup 5 --cpu-seconds 8 > gpurun_out/bench.log 2>&1
rc=$?
grep '^{' gpurun_out/bench.log
exit $rc
