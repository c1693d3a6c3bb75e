#!/bin/bash
# usage: tools/build_variants.sh name "EXTRA flags" [name "flags" ...]  -> tools/variants/<name>.so
# Only the kernel objects are rebuilt per variant (fused.o, bigfused_*.o, kernels.o); ALL=1 rebuilds everything.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/stabilizer-stream_amd/csrc
mkdir -p $root/tools/variants
make -j6 -C $csrc >/dev/null
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  if [ -n "$ALL" ]; then make -C $csrc clean >/dev/null; else rm -f $csrc/fused.o ${BIG:+$csrc/bigfused*.o}; fi
  make -j6 -C $csrc EXTRA="$flags" >/dev/null
  cp $root/stabilizer-stream_amd/libpsdcascade.so $root/tools/variants/$name.so
  echo built $name
done
if [ -n "$ALL" ]; then make -C $csrc clean >/dev/null; else rm -f $csrc/fused.o ${BIG:+$csrc/bigfused*.o}; fi
make -j6 -C $csrc >/dev/null
echo restored default build
