"""usage: python tools/chk_bench_line.py [file with a bench.py JSON line]  (default: gpurun_out/bench/bench.log)
Prints the fields the round's record quotes: headline, roofline (binding resource, counter traffic, device state), the scalar-FMA
compute roofline, other_configs with their Loss, hbm_honest, cpu_baseline."""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench/bench.log"
d = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
r, c = d["roofline"], d["compute_roofline"]
print(f"{d['value']:.0f} {d['unit']}, {d['ms_per_step']:.4f} ms/step, {d['config']['workload'][:120]}")
print("roofline", {k: r.get(k) for k in ("bound", "frac", "achieved", "avg_launch_ms", "launches", "traffic_over_algorithmic",
                                          "counter_traffic_frac", "traffic_source")})
print("binding:", r.get("binding"))
print("device_state:", r.get("device_state"))
print("compute", {k: c.get(k) for k in ("frac", "peak", "scalar_fma_peak", "frac_of_scalar_fma_peak")})
for k, v in (d.get("other_configs") or {}).items():
    print(k, f"{v['value']:.0f}", f"frac {v['roofline']['frac']:.4f}", "loss", v.get("loss"))
if d.get("hbm_honest"):
    print("hbm_honest", f"{d['hbm_honest']['value']:.0f}", f"frac {d['hbm_honest']['roofline']['frac']:.4f}")
if d.get("cpu_baseline"):
    cb = d["cpu_baseline"]
    print("cpu_baseline", {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "scalar_radix2_fft_value", "n512_value", "host")})
if d.get("host_fed"):
    print("host_fed", f"{d['host_fed']['value']:.0f} MS/s")
