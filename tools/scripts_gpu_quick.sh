#!/bin/bash
# quick: N=1024 parity subset + bench
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "1024 or golden or device or full_size or statistical" > gpurun_out/pytest_quick.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_quick.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
rc=$?
grep '^{' gpurun_out/bench_quick.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'host enqueue ms/step',round(d['host_enqueue_ms_per_step'],4),'kernel avg ms',round(d['roofline']['avg_launch_ms'],4),'launches',d['roofline']['launches'])"
exit $rc
