"""Prints the headline and the per-kernel lines of a bench.py JSON line."""
import json
import sys
d = json.load(open(sys.argv[1]))
print(f"{d['ms_per_step']:.3f} ms/step  {d['value']:.0f} {d['unit']}  roofline: {d.get('roofline')}")
for k, v in d.get("kernels", {}).items():
    print(f"  {k:60s} {v['ms'] * 1e3:8.1f} us  frac {v['frac']:.3f}  x{v['launches_per_step']}")
