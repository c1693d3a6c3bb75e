#!/bin/bash
# host-side feeds: parity tests that touch them, then the frame path and the default bench (host_fed field)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "adcdac or frames or source or chunk or stress or host" > gpurun_out/pytest_host.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_host.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_frames.py 2>&1 | tail -2
timeout -k 10 600 python bench.py --cpu-seconds 2 > gpurun_out/bench_host.log 2>&1
rc=$?
grep '^{' gpurun_out/bench_host.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s',round(d['value']),'host_fed',d['host_fed'])"
exit $rc
