#!/bin/bash
# compare the default library and every tools/variants/*.so on quoted bench argument strings
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  for so in default tools/variants/*.so; do
    if [ "$so" = default ]; then unset PSDC_LIB; else export PSDC_LIB=$GRAFT_REPO_ROOT/$so; fi
    python bench.py $a --no-cpu-baseline --no-other-configs 2>>gpurun_out/cmp_err.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$a |', '$so', 'MS/s', round(d['value']), 'kernel frac', round(r['frac'],4), 'avg launch ms', round(r['avg_launch_ms'],4))"
  done
done
