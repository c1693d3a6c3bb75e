#!/bin/bash
# first GPU pass: parity tests, smoke, short bench. Stops if any step is killed by its timeout.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc2=$?
echo "smoke rc=$rc2"; tail -5 gpurun_out/smoke.log
if [ $rc2 -eq 124 ] || [ $rc2 -eq 137 ]; then exit $rc2; fi
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 5 > gpurun_out/bench_first.log 2>&1
rc3=$?
echo "bench rc=$rc3"; tail -3 gpurun_out/bench_first.log
exit $rc
