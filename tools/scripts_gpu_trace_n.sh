#!/bin/bash
# kernel timeline of a short bench at the given bench.py args
mkdir -p gpurun_out/tr
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -o t -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tr/bench.log 2>&1
python3 - <<'PY'
import csv,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/tr/'
rows=list(csv.DictReader(open(root+'t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=[i for i,r in enumerate(rows) if 'fused' in r['Kernel_Name'] and int(r['End_Timestamp'])-int(r['Start_Timestamp'])>50000]
i0=max(0,big[-3]-2)
prev=None
for r in rows[i0:i0+12]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"gap {((s-prev)/1e3 if prev else 0):7.1f}  dur {(e-s)/1e3:8.1f}us  grid={r['Grid_Size_X']:>9} wg={r['Workgroup_Size_X']:>4} {r['Kernel_Name'][:48]}")
    prev=e
PY
