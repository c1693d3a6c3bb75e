#!/bin/bash
# usage: tools/build_variants.sh name "EXTRA flags" [name "flags" ...]  -> tools/variants/<name>.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/stabilizer-stream_amd/csrc
mkdir -p $root/tools/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  make -C $csrc clean >/dev/null
  make -j6 -C $csrc EXTRA="$flags" >/dev/null
  cp $root/stabilizer-stream_amd/libpsdcascade.so $root/tools/variants/$name.so
  echo built $name
done
make -C $csrc clean >/dev/null
make -j6 -C $csrc >/dev/null
echo restored default build
