#!/bin/bash
mkdir -p gpurun_out/mc
python bench.py --n 1024 --channels-per-gpu 8 --log2-batch 23 --steps 60 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MS/s',round(d['value']),'ms/step',round(d['ms_per_step'],4),'host ms/step',round(d['host_enqueue_ms_per_step'],4),'kernel avg ms',round(d['roofline']['avg_launch_ms'],4),'launches',d['roofline']['launches'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mc -o t -- python3 $GRAFT_REPO_ROOT/bench.py --n 1024 --channels-per-gpu 8 --log2-batch 23 --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/mc/bench.log 2>&1
python3 - <<'PY'
import csv,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/mc/'
rows=list(csv.DictReader(open(root+'t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=[i for i,r in enumerate(rows) if 'fused' in r['Kernel_Name'] and int(r['End_Timestamp'])-int(r['Start_Timestamp'])>50000]
i0=big[-4]-2
base=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+14]:
    s=int(r['Start_Timestamp'])-base; e=int(r['End_Timestamp'])-base
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}us  grid={r['Grid_Size_X']:>9} wg={r['Workgroup_Size_X']:>4} {r['Kernel_Name'][:50]}")
PY
