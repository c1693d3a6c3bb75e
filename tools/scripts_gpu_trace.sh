#!/bin/bash
# kernel trace + SQ stall counters of the dominant kernel for the current build
mkdir -p gpurun_out/tr
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tr/bench.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tr/bench_sq.log 2>&1 || exit 2
python3 - <<'PY'
import csv,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/tr/'
rows=list(csv.DictReader(open(root+'t_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])): print(f"{k[:50]:50s} n={len(v):3d} max={max(v)/1e3:8.1f}us med={sorted(v)[len(v)//2]/1e3:8.1f}us tot={sum(v)/1e3:9.1f}us")
rows=list(csv.DictReader(open(root+'sq_counter_collection.csv')))
by=collections.defaultdict(dict)
for r in rows:
    if 'fused_kernel' in r['Kernel_Name']:
        by[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value']); by[r['Dispatch_Id']]['_dur']=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
if by:
    b=max(by.values(), key=lambda x:x['_dur'])
    wc=b['SQ_WAVE_CYCLES']
    print({k:(round(v/wc,3) if k.startswith('SQ_') and k not in ('SQ_WAVE_CYCLES',) else v) for k,v in b.items()})
PY
