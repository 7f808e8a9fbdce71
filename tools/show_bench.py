#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line."""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  n_gpus={d['n_gpus']}")
r = d.get("roofline")
if r:
    print(f"conv family: {r['achieved']:.1f} TFLOP/s = {r['frac']*100:.2f}% of {r['peak']:.0f}; device ms/step all kernels {r['device_ms_per_step_all_kernels']:.3f}")
    for k, v in r["by_kernel"].items():
        print(f"  {k:16s} {v['launches_per_step']:5.1f} launches {v['ms_per_step']:8.4f} ms/step {v['avg_us_per_launch']:8.1f} us  tflops={v['tflops']}  GB/s={v['gbs']}")
if "cpu_baseline" in d:
    print("cpu_baseline:", d["cpu_baseline"])
