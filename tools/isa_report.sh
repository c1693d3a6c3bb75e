#!/bin/bash
# usage: tools/isa_report.sh [file.hip ...]   (default: every fused kernel TU)
# Compiles the kernel TUs to gfx950 assembly with the Makefile's flags (device side only, no GPU needed) and prints per kernel:
# VGPRs, SGPRs, scratch bytes and how many scratch loads / stores sit INSIDE a loop (a reload inside the pair loop waits on
# vmcnt behind the look-ahead loads: the N = 16384 Mean kernel lost 24 % to thirteen of them).
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/stabilizer-stream_amd/csrc
out=${ISA_OUT:-/tmp/isa_report}
mkdir -p $out
files="$@"
[ -z "$files" ] && files="fused.hip bigfused_2048.hip bigfused_4096.hip bigfused_8192.hip bigfused_16384.hip bigfused3_2048.hip bigfused3_4096.hip"
for f in $files; do
  sched=""
  [ "$f" = fused.hip ] && sched="-mllvm -amdgpu-sched-strategy=max-ilp"
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $sched \
      -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA -I$csrc --cuda-device-only -S $csrc/$f -o $out/${f%.hip}.s 2>/dev/null
  python3 - $out/${f%.hip}.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)s_endpgm', s, re.M | re.S):
    name, body = m.group(1), m.group(2)
    tail = s[m.end():m.end() + 6000]
    g = lambda k: (re.search(r'; %s: (\d+)' % k, tail) or [None, '?'])[1]
    inloop = depth = 0
    loads = stores = lin = sin = 0
    cur_in_loop = False
    for line in body.split('\n'):
        if line.startswith('.LBB'):
            cur_in_loop = 'in Loop' in line or 'Loop Header' in line
        if 'scratch_load' in line:
            loads += 1
            lin += cur_in_loop
        if 'scratch_store' in line:
            stores += 1
            sin += cur_in_loop
    short = re.sub(r'^_ZN4psdk\d+', '', name)
    short = re.sub(r'EvNS_.*$', '', short)
    print(f"{short:46s} vgpr {g('NumVgprs'):>3s} sgpr {g('TotalNumSgprs'):>3s} scratch {g('ScratchSize'):>4s} B  reloads {loads:3d} ({lin} in loops)  spills {stores:3d} ({sin} in loops)")
PY
done
