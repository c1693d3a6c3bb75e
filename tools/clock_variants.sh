#!/bin/bash
# engine clock / socket power of every tools/variants/*.so (timing-only ablations) against the default build: what each part of
# the kernel costs in POWER, the resource the headline kernel runs out of (tools/clock_probe.sh)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/clock
for so in default tools/variants/*.so; do
  if [ "$so" = default ]; then unset PSDC_LIB; else export PSDC_LIB=$PWD/$so; fi
  python bench.py --steps 5000 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/clock/bench.log 2>&1 &
  pid=$!
  sleep 6
  s=""; n=0
  while kill -0 $pid 2>/dev/null && [ $n -lt 6 ]; do
    s="$s $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Package Power' | sed 's/.*: //' | tr '\n' ' ')"
    n=$((n+1)); sleep 0.4
  done
  wait $pid
  echo "$so | $(grep '^{' gpurun_out/clock/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s', round(d['value']))") | $s"
done
