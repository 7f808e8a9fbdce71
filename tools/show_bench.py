"""Pretty-print bench.py JSON lines: python tools/show_bench.py gpurun_out/x.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
        continue
    r = d.pop("roofline", None)
    print(f"{f}: {d['value']:.1f} {d['unit']}  {d['ms_per_step']:.4f} ms/step  n_gpus={d['n_gpus']} ranks={d.get('ranks')} "
          f"backend={d.get('backend')}")
    for k in ("cold", "group1", "mixed_scales", "offline", "dp_strict", "replicas", "infer", "cpu_baseline"):
        if k in d:
            print("  ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d[k].items() if a not in ("protocol", "sample", "parallelism", "note", "workload")})
    if r:
        bk = r.pop("by_kernel")
        print("   roofline", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                              if k not in ("traffic_note", "kernel", "measured")})
        for k, v in bk.items():
            print(f"     {k[:58]:58s} n={v['launches_per_step']:5.1f} ms/step={v['ms_per_step']:.4f} avg_us={v['avg_us_per_launch']:7.2f} tflops={v['tflops']}")
