#!/bin/bash
# SQ counter passes for the dominant kernel (own runs; kernel-trace only besides --pmc)
mkdir -p gpurun_out/sq
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq -o sq$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/sq/bench_$i.log 2>&1
  rc=$?
  echo "set $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $GRAFT_REPO_ROOT/gpurun_out/sq/bench_$i.log; fi
done
ls $GRAFT_REPO_ROOT/gpurun_out/sq
