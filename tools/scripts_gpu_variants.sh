#!/bin/bash
# bench the prebuilt library variants in tools/variants/ (tools/build_variants.sh); args = bench.py args
mkdir -p gpurun_out
for so in tools/variants/*.so; do
  cp $so stabilizer-stream_amd/libpsdcascade.so
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline > gpurun_out/bench_var.log 2>&1
  rc=$?
  echo "== $so rc=$rc"
  grep '^{' gpurun_out/bench_var.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'kernel avg ms',round(d['roofline']['avg_launch_ms'],4),'launches',d['roofline']['launches'])" || tail -3 gpurun_out/bench_var.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
