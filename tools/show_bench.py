import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.0f} tok/s  {d['ms_per_step']:.1f} ms/step  n_gpus={d['n_gpus']}")
for k, v in d.get("roofline_all", {}).items():
    print(f"  {k:64s} {v['achieved']:8.1f} {v['unit']:8s} frac {v['frac']:.3f}  avg {v['avg_ms']*1e3:7.1f} us  x{v['launches']:4d}  {v['total_ms']/d.get('roofline_steps', d['steps']):6.1f} ms/step")
if "cpu_baseline" in d:
    print("  cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
