#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1
rc=$?
grep '^{' gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','host_fed','cpu_baseline')}, indent=0)); print(d['roofline'])"
exit $rc
