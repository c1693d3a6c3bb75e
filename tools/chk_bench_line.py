import json,sys
d=json.loads(open("gpurun_out/final/driver_style.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic_source"))
print(json.dumps(d["other_configs"])[:1800])
print(json.dumps(d["cpu_baseline"])[:700])
print(json.dumps(d.get("hbm_honest")))
