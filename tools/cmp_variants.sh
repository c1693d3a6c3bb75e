#!/bin/bash
# compare variants: for each N run default lib and each tools/variants/*.so
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  for so in default tools/variants/*.so; do
    if [ "$so" = default ]; then unset PSDC_LIB; else export PSDC_LIB=$GRAFT_REPO_ROOT/$so; fi
    python bench.py --n $n --steps 60 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('N=$n', '$so', 'MS/s', round(d['value']), 'kernel frac', round(r['frac'],4), 'avg launch ms', round(r['avg_launch_ms'],4))"
  done
done
