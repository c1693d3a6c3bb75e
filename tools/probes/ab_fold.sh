set -e
cd tests/host
for v in default back wg backwg; do
  d=/tmp/v_$v; mkdir -p $d
  if [ $v = default ]; then cp ../../stabilizer-stream_amd/libpsdcascade.so $d/; else cp ../../tools/variants/$v.so $d/libpsdcascade.so; fi
done
for rep in 1 2; do
for v in default back wg backwg nofold; do
  d=/tmp/v_$v; e=""
  if [ $v = nofold ]; then d=/tmp/v_default; export PSDC_NO_FOLD=1; else unset PSDC_NO_FOLD; fi
  for lg in 16 18 20 22; do
    echo "$v $lg $(LD_LIBRARY_PATH=$d ./devcall_probe 1024 0.3 0 0 $lg | python3 -c "import sys,json; j=json.load(sys.stdin); print({k:(v2['with_drain']//1000) for k,v2 in j['scattered'].items()})")"
  done
done
done
unset PSDC_NO_FOLD
cd ../..
for rep in 1 2; do
for v in default back wg backwg nofold; do
  lib=/tmp/v_$v/libpsdcascade.so
  if [ $v = nofold ]; then lib=/tmp/v_default/libpsdcascade.so; export PSDC_NO_FOLD=1; else unset PSDC_NO_FOLD; fi
  for cfg in "" "--channels-per-gpu 8 --log2-batch 24"; do
    echo "$v [$cfg] $(PSDC_LIB=$lib python bench.py --no-cpu-baseline --no-other-configs --steps 100 $cfg | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']/1000,1), round(j['roofline']['frac'],4))")"
  done
done
done
