#!/bin/bash
# tools/probes/ab_libs.sh "<bench args>" name name ... -- run ON the GPU box: bench.py with the given arguments once per NAME, in the
# order given (write the order as A B B A ...: the second of two back-to-back bench processes reads ~1 % high whichever it is).
# NAME = default | a variant in tools/variants/NAME.so (loaded through PSDC_LIB) | NAME+ENV=VALUE (default library with that
# environment variable, e.g. default+PSDC_NO_FOLD=1).
args=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root"
for name in "$@"; do
  lib=${name%%+*}; env=""; [ "$lib" != "$name" ] && env=${name#*+}
  so=""; [ "$lib" != default ] && so="$root/tools/variants/$lib.so"
  echo "$name $(env ${env:+$env} ${so:+PSDC_LIB=$so} python bench.py --no-cpu-baseline --no-other-configs $args 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']/1000,1), 'GS/s, kernel-only', round(j['roofline']['frac'],4), 'launches', j['roofline'].get('launches'))")"
done
