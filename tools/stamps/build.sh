#!/bin/bash
# Instrumented build of the fused kernel (phase stamps, -DPSDK_STAMPS) into
# tools/stamps/libpsdcascade_stamps.so.  The shipped library is never built this way.
set -e
here=$(cd "$(dirname "$0")" && pwd)
csrc=$here/../../stabilizer-stream_amd/csrc
make -j4 -C "$csrc" >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DPSDK_STAMPS ${STAMP_EXTRA} \
    -I"$csrc" -c "$csrc/fused.hip" -o "$here/fused_stamps.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$here/libpsdcascade_stamps.so" "$csrc/kernels.o" \
    "$here/fused_stamps.o" "$csrc"/bigfused_*.o "$csrc/psdcascade.o"
echo built "$here/libpsdcascade_stamps.so"
