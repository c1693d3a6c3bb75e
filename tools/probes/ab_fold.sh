#!/bin/bash
# tools/probes/ab_fold.sh [reps = 2] -- run ON the GPU box (gpurun -- bash tools/probes/ab_fold.sh > gpurun_out/ab_fold.log):
# scattered device calls (tests/host/devcall_probe, 2^16 ... 2^22 samples a call) and bench.py (1 and 8 channels), alternating
# between the default library, every tools/variants/*.so (tools/build_variants.sh) and the default under PSDC_NO_FOLD=1 (every
# round through post_kernel).  Each library sits in a directory of its own and is the only one its process loads
# (LD_LIBRARY_PATH for the probe, PSDC_LIB for the Python package): never LD_PRELOAD a variant over the default.
set -e
reps=${1:-2}
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root/tests/host"
names="default"
mkdir -p /tmp/v_default && cp "$root/stabilizer-stream_amd/libpsdcascade.so" /tmp/v_default/
for so in "$root"/tools/variants/*.so; do
  [ -e "$so" ] || continue
  v=$(basename "$so" .so); names="$names $v"
  mkdir -p /tmp/v_$v && cp "$so" /tmp/v_$v/libpsdcascade.so
done
names="$names nofold"
pick() { if [ "$1" = nofold ]; then d=/tmp/v_default; export PSDC_NO_FOLD=1; else d=/tmp/v_$1; unset PSDC_NO_FOLD; fi; }
for rep in $(seq $reps); do
  for v in $names; do
    pick $v
    for lg in 16 18 20 22; do
      echo "$v 2^$lg scattered GS/s: $(LD_LIBRARY_PATH=$d ./devcall_probe 1024 0.3 0 0 $lg | python3 -c "import sys,json; j=json.load(sys.stdin); print([v2['with_drain']//1000 for v2 in j['scattered'].values()][0])")"
    done
  done
done
cd "$root"
for rep in $(seq $reps); do
  for v in $names; do
    pick $v
    for cfg in "" "--channels-per-gpu 8 --log2-batch 24"; do
      echo "$v bench [$cfg] $(PSDC_LIB=$d/libpsdcascade.so python bench.py --no-cpu-baseline --no-other-configs --steps 100 $cfg 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']/1000,1), 'GS/s, kernel-only', round(j['roofline']['frac'],4))")"
    done
  done
done
