#!/bin/bash
# rehearsal of the multi-rank bench path on ONE GPU: 2 ranks, gloo, both on cuda:0
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo --single-device --log2-batch 24 > gpurun_out/bench_mr.log 2>&1
rc=$?
echo "rc=$rc"; grep '^{' gpurun_out/bench_mr.log | cut -c1-600 || tail -20 gpurun_out/bench_mr.log
[ $rc -eq 0 ] || tail -20 gpurun_out/bench_mr.log
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --cpu-seconds 6 > gpurun_out/bench_default.log 2>&1
echo "rc=$?"; grep '^{' gpurun_out/bench_default.log
# the RCCL calls of the N > 1 path (init with device_id, barrier, device-tensor gather, all_reduce) in a 1-rank group
PSD_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_rccl1.log 2>&1
echo "rccl 1-rank rc=$?"; grep '^{' gpurun_out/bench_rccl1.log | cut -c1-300 || tail -20 gpurun_out/bench_rccl1.log
