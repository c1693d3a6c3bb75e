#!/bin/bash
# tools/clock_probe.sh [bench args]: engine clock and package power (rocm-smi) sampled while bench.py runs -- the fused kernels run
# power-limited (GRBM_GUI_ACTIVE / duration reads ~1.8 GHz of the 2.4 GHz peak), so per-clock efficiency and GS/s are two things.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/clock
python bench.py --steps 6000 --no-cpu-baseline --no-other-configs "$@" > gpurun_out/clock/bench.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.4
done
wait $pid
grep '^{' gpurun_out/clock/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s', round(d['value']), 'frac', round(d['roofline']['frac'],4))"
