#!/bin/bash
# Instrumented build of the fused kernel (phase stamps, -DPSDK_STAMPS) into
# tools/stamps/libpsdcascade_stamps.so.  The shipped library is never built this way.
set -e
here=$(cd "$(dirname "$0")" && pwd)
csrc=$here/../../stabilizer-stream_amd/csrc
make -j4 -C "$csrc" >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
    -Xclang -target-feature -Xclang -packed-fp32-ops -DPSDK_STAMPS ${STAMP_EXTRA} \
    -I"$csrc" -c "$csrc/fused.hip" -o "$here/fused_stamps.o" 2> >(grep -v packed-fp32 >&2)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$here/libpsdcascade_stamps.so" "$csrc/kernels.o" "$csrc/bigfft.o" \
    "$here/fused_stamps.o" "$csrc"/bigfused_*.o "$csrc"/bigfused3_*.o "$csrc/runtime.o" "$csrc/planner.o" "$csrc/frames_ingest.o" "$csrc/readout.o"
echo built "$here/libpsdcascade_stamps.so"
