#!/bin/bash
# Instrumented build of ONE workgroup-level fused kernel (phase stamps, -DPSDK_STAMPS): tools/stamps/build_big.sh 4096
# -> tools/stamps/libpsdcascade_bstamps_<N>.so.  The shipped library is never built this way.
set -e
n=${1:-4096}
here=$(cd "$(dirname "$0")" && pwd)
csrc=$here/../../stabilizer-stream_amd/csrc
make -j4 -C "$csrc" >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -DPSDK_STAMPS ${STAMP_EXTRA} \
    -I"$csrc" -c "$csrc/bigfused_$n.hip" -o "$here/bigfused_${n}_stamps.o"
objs=""
for m in 2048 4096 8192 16384; do
  if [ "$m" = "$n" ]; then objs="$objs $here/bigfused_${n}_stamps.o"; else objs="$objs $csrc/bigfused_$m.o"; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$here/libpsdcascade_bstamps_$n.so" "$csrc/kernels.o" "$csrc/bigfft.o" "$csrc/fused.o" $objs "$csrc/bigfused3_2048.o" "$csrc/bigfused3_4096.o" \
    "$csrc/runtime.o" "$csrc/planner.o" "$csrc/frames_ingest.o" "$csrc/readout.o"
echo built "$here/libpsdcascade_bstamps_$n.so"
