#!/bin/bash
# tools/gpu.sh MODE [args] -- the developer passes on the 1-GPU box, one script:
#   gpurun --timeout 900 -- ./tools/gpu.sh tests [-k expr]
# Everything lands under gpurun_out/<mode>/.  A step killed by its timeout ends the call (no further GPU step).
#
#   tests [pytest args]     pytest -m gpu
#   allvariants [args]      pytest -m gpu under PSDC_DBG_VARIANT = 0, 1, 2, 3 (every fused launch on the EWMA / FRAMES kernel variants)
#   smoke                   __graft_entry__.smoke()
#   bench [bench args]      one bench.py line, condensed
#   lines "<args>" ...      one condensed bench line per quoted argument string
#   shapes                  the standard table: every N, detrends, EWMA, multi-channel
#   stats TAG [bench args]  rocprofv3 --kernel-trace --stats of bench.py  -> gpurun_out/stats/TAG_*
#   pmc TAG [bench args]    FETCH_SIZE / WRITE_SIZE passes (own runs)      -> gpurun_out/pmc/TAG_*
#   sq TAG [bench args]     SQ counter passes of the dominant kernel       -> gpurun_out/sq/TAG_*
#   timeline [bench args]   kernel timeline around the last dominant launches (gaps between kernels)
#   record TAG              tests + default bench + stats + pmc: the record of a build (make_profile_summary.py TAG)
#   variants [bench args]   bench every tools/variants/*.so (tools/build_variants.sh)
#   ranks N [bench args]    bench.py --gpus N --backend gloo --single-device (N ranks on the one GPU)
#   py FILE [args]          run a tools/*.py helper
mode=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/$mode
mkdir -p "$out"
cd "$R" || exit 1
killed() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
condense='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "| MS/s", round(d["value"]), "ms/step", round(d["ms_per_step"],4), "host ms/step", round(d.get("host_enqueue_ms_per_step",0),4),
      "kernel avg ms", round(r["avg_launch_ms"],4), "launches", r["launches"], "frac", round(r["frac"],3), "e2e_frac", round(4e-3*d["value"]/8000/max(1,d["n_gpus"]),3),
      d["config"]["stages"], "stages")'
line() { # line "<bench args>" [extra args]
  local a="$1"; shift
  timeout -k 10 400 python bench.py $a "$@" --no-cpu-baseline --no-other-configs > "$out/bench.log" 2>&1; local rc=$?
  grep '^{' "$out/bench.log" | python -c "$condense" "$a" || { echo "$a rc=$rc"; tail -5 "$out/bench.log"; }
  grep '^{' "$out/bench.log" >> "$out/lines.jsonl"
  return $rc
}
prof() { # prof <subdir> <prefix> <rocprof flags...> -- <bench args>
  local o="$R/gpurun_out/$1" p=$2; shift 2
  local flags=(); while [ "$1" != "--" ]; do flags+=("$1"); shift; done; shift
  mkdir -p "$o"
  (cd /tmp && TMPDIR=/tmp timeout -k 10 700 rocprofv3 "${flags[@]}" --output-format csv -d "$o" -o "$p" -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-other-configs > "$o/${p}_bench.log" 2>&1)
}
case $mode in
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 "$@" > "$out/pytest.log" 2>&1; rc=$?
  echo "pytest rc=$rc"; tail -8 "$out/pytest.log"; exit $rc ;;
allvariants)
  # the whole GPU suite on each of the four kernel variants (PSDC_DBG_VARIANT: bit 0 EWMA, bit 1 FRAMES kernels for every fused launch)
  for v in 0 1 2 3; do
    PSDC_DBG_VARIANT=$v timeout -k 10 400 python -m pytest tests -m gpu -q -x --timeout 900 "$@" > "$out/pytest_v$v.log" 2>&1; rc=$?
    echo "PSDC_DBG_VARIANT=$v rc=$rc: $(tail -1 "$out/pytest_v$v.log")"
    if [ $rc -ne 0 ]; then tail -8 "$out/pytest_v$v.log"; exit $rc; fi
  done ;;
smoke)
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
bench)
  timeout -k 10 900 python bench.py "$@" > "$out/bench.log" 2>&1; rc=$?
  grep '^{' "$out/bench.log" | python -c "$condense" "$*" || tail -20 "$out/bench.log"
  grep '^{' "$out/bench.log" | cut -c1-1500; exit $rc ;;
lines)
  for a in "$@"; do line "$a"; rc=$?; if killed $rc; then exit $rc; fi; done ;;
shapes)
  for a in "--n 256" "--n 512" "--n 1024" "--n 2048" "--n 4096" "--n 8192" "--n 16384" \
           "--n 1024 --detrend midpoint" "--n 1024 --detrend span" "--n 1024 --detrend mean" "--n 4096 --detrend mean" \
           "--n 1024 --avg 1000,100000" "--n 1024 --channels-per-gpu 2 --log2-batch 25" \
           "--n 1024 --channels-per-gpu 8 --log2-batch 24" "--n 1024 --channels-per-gpu 8 --log2-batch 23" \
           "--n 64" "--n 128"; do
    line "$a" "$@"; rc=$?; if killed $rc; then exit $rc; fi
  done ;;
stats)
  tag=$1; shift
  prof stats "$tag" --kernel-trace --stats -- "$@"; rc=$?
  echo "stats rc=$rc"; cut -c1-170 "$R/gpurun_out/stats/${tag}_kernel_stats.csv" | head -12
  grep '^{' "$R/gpurun_out/stats/${tag}_bench.log" | python -c "$condense" "$*"; exit $rc ;;
pmc)
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    prof pmc "${tag}_$c" --pmc $c --kernel-trace -- --steps 5 --warmup 2 --coalesce 1 --passes 1 "$@"; rc=$?
    echo "$c rc=$rc"; if [ $rc -ne 0 ]; then tail -5 "$R/gpurun_out/pmc/${tag}_${c}_bench.log"; exit $rc; fi
  done
  python3 tools/make_profile_summary.py --traffic-only "$tag" ;;
sq)
  tag=$1; shift
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
             "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    prof sq "${tag}_set$i" --pmc $set --kernel-trace -- --steps 3 --warmup 1 --passes 1 "$@"; rc=$?
    echo "set $i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 "$R/gpurun_out/sq/${tag}_set${i}_bench.log"; exit $rc; fi
  done
  python3 tools/make_profile_summary.py --sq-only "$tag" ;;
timeline)
  prof timeline t --kernel-trace -- --steps 6 --warmup 2 "$@" || exit $?
  python3 - <<'PY'
import csv, os
root = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/timeline/'
rows = sorted(csv.DictReader(open(root + 't_kernel_trace.csv')), key=lambda r: int(r['Start_Timestamp']))
big = [i for i, r in enumerate(rows) if 'fused' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 50000]
i0 = max(0, big[-3] - 2) if len(big) >= 3 else 0
prev = None
for r in rows[i0:i0 + 14]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"gap {((s - prev) / 1e3 if prev else 0):7.1f}  dur {(e - s) / 1e3:8.1f}us  grid={r['Grid_Size_X']:>9} wg={r['Workgroup_Size_X']:>4} {r['Kernel_Name'][:48]}")
    prev = e
PY
  ;;
record)
  tag=$1; shift
  if [ -n "$PSDC_LIB" ]; then echo "record: PSDC_LIB is set ($PSDC_LIB): a record is of the default build only"; exit 2; fi
  "$0" tests || exit $?
  "$0" bench || exit $?
  "$0" stats "$tag" || exit $?
  "$0" pmc "$tag" || exit $?
  echo "record $tag complete: python tools/make_profile_summary.py $tag" ;;
variants)
  # each variant is loaded through $PSDC_LIB: the shipped stabilizer-stream_amd/libpsdcascade.so is never overwritten
  # (the PSDK_ABL builds give wrong results on purpose; a later tests / bench / record must not run against one)
  for so in tools/variants/*.so; do
    PSDC_LIB="$R/$so" line "$*"; rc=$?; echo "   ^ $so"; if killed $rc; then exit $rc; fi
  done ;;
ranks)
  n=$1; shift
  timeout -k 10 600 python bench.py --gpus "$n" --backend gloo --single-device "$@" > "$out/bench.log" 2>&1; rc=$?
  echo "rc=$rc"; grep '^{' "$out/bench.log" | cut -c1-900 || tail -20 "$out/bench.log"; exit $rc ;;
py)
  f=$1; shift
  timeout -k 10 1100 python "$f" "$@" ;;
*)
  echo "unknown mode $mode"; sed -n 2,20p "$0"; exit 2 ;;
esac
