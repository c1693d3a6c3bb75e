#!/bin/bash
# tools/planner_mutations.sh -- does tests/host/round_plan_check notice a broken planner?  Six one-line bugs are seeded into a
# COPY of the host runtime (csrc/planner.cpp, csrc/runtime.cpp) (a tail carried one sample short; the fused jobs' output pointer off by one; a seam four samples
# short; stream buffers grown one float too small; a seam copy riding as the prologue of a job of TWO workgroups; a one-launch round
# folding the partial slab it writes) and the check is rebuilt against each: every one must fail (identity mismatches or the
# model's own complaints for five of them, an AddressSanitizer heap-buffer-overflow for the fourth).  CPU only, ~8 minutes (sixty seeds: the last samples of the head-of-span seam copy of mutation 3 are read by few scenarios -- first caught at seed 41 of the round-5 scenario stream).
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
C=$root/stabilizer-stream_amd/csrc
H=$root/tests/host
T=${TMPDIR:-/tmp}/psdc_mut
mkdir -p $T
bad=0
for m in 1 2 3 4 5 6; do
  cp $C/runtime.cpp $C/planner.cpp $C/frames_ingest.cpp $C/readout.cpp $C/host_runtime.h $T/
  cat $C/planner.cpp $C/runtime.cpp > $T/before.txt
  case $m in
    1) sed -i 's/const uint64_t cnt = told > kf ? told - kf : 0;/const uint64_t cnt = told > kf + 1 ? told - kf - 1 : 0;/' $T/planner.cpp ;;
    2) sed -i 's/fj.dst = nx->buf.p\[nx->buf.cur ^ 1\] + (mf0 - g.drain - nx_base);/fj.dst = nx->buf.p[nx->buf.cur ^ 1] + (mf0 - g.drain - nx_base) + 1;/' $T/planner.cpp ;;
    3) sed -i 's/buf + (c.spans\[0\].first - s0.buf.base), (size_t)cp0));/buf + (c.spans[0].first - s0.buf.base), (size_t)(cp0 > 4 ? cp0 - 4 : cp0)));/' $T/planner.cpp ;;
    5) sed -i 's/if (pf.j.nblocks != 1)/if (pf.j.nblocks > 2)/' $T/planner.cpp ;;
    6) sed -i 's/^    h->partial_cur ^= 1;/    h->partial_cur ^= 0;/' $T/planner.cpp ;;
    4) sed -i 's/const size_t min_cap = (size_t)4 \* (h->n + HBF_HALO) + 64;/const size_t min_cap = 16;/; s/size_t cap = std::max(need + need \/ 2, min_cap);/size_t cap = std::max(need - 1, min_cap);/' $T/runtime.cpp ;;
  esac
  if cat $T/planner.cpp $T/runtime.cpp | cmp -s - $T/before.txt; then echo "mutation $m did not apply (the source moved on: update this script)"; bad=1; continue; fi
  sed -i "s#\"../../include/psdcascade.h\"#\"$root/include/psdcascade.h\"#" $T/host_runtime.h
  g++ -O1 -g -std=c++17 -w -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$H/sim -I$T -I$C -I$H $H/round_plan_check.cpp $H/sim/sim_kernels.cpp \
      $T/runtime.cpp $T/planner.cpp $T/frames_ingest.cpp $T/readout.cpp -o $T/chk$m -lpthread || { echo "mutation $m: build failed"; bad=1; continue; }
  if timeout 1500 $T/chk$m 1 60 > $T/out$m.log 2>&1; then echo "mutation $m: NOT DETECTED"; bad=1
  else echo "mutation $m: detected ($(grep -c '^FAIL' $T/out$m.log) failures, $(grep -c 'ERROR: AddressSanitizer' $T/out$m.log) ASan report)"; fi
done
exit $bad
