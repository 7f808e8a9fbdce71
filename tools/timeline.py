#!/usr/bin/env python3
"""Timeline of one accumulation cycle from a rocprofv3 --kernel-trace CSV: per HIP queue, the kernels in start order with
gaps, between two optimizer steps (k_sgd).   usage: timeline.py trace.csv [cycle-index-from-end] [--overlap]
(trace.csv: rocprofv3's kernel_trace.csv, or the launch-event file a lab build writes under FOSVOS_PROF_TIMELINE=<file>)
--overlap: instead of the kernel list, how the cycle's wall time splits by the number of kernels in flight, and which kernels
run with nothing beside them (time-weighted)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:44]


def family(n):
    for p in ("k_conv3x3_pp", "k_conv3x3_igemm", "k_wgrad3x3_v2", "k_wgrad_first"):
        if p in n:
            return p
    return "other"


def overlap_report(ks, t0, t1):
    """Sweep over the start / end times inside [t0, t1)."""
    ev = []
    for i, k in enumerate(ks):
        s, e = max(k["s"], t0), min(k["e"], t1)
        if s < e:
            ev.append((s, 1, i))
            ev.append((e, 0, i))
    ev.sort()
    active, last = set(), t0
    by_count, solo, pairs = {}, {}, {}
    for t, kind, i in ev:
        if t > last:
            n = len(active)
            by_count[n] = by_count.get(n, 0) + t - last
            if n == 1:
                nm = ks[next(iter(active))]["n"]
                solo[nm] = solo.get(nm, 0) + t - last
            if n >= 1:
                key = "+".join(sorted(family(ks[j]["n"]) for j in active))
                pairs[key] = pairs.get(key, 0) + t - last
            last = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    wall = t1 - t0
    by_count[0] = by_count.get(0, 0) + max(0, t1 - last)
    print("kernels in flight: " + "  ".join(f"{n}: {100 * v / wall:.1f}%" for n, v in sorted(by_count.items())))
    print("what runs together (share of the cycle):")
    for key, v in sorted(pairs.items(), key=lambda kv: -kv[1])[:16]:
        print(f"  {100 * v / wall:5.1f}%  {key}")
    print("alone in flight (share of the cycle):")
    for nm, v in sorted(solo.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {100 * v / wall:5.1f}%  {nm}")


def main():
    overlap = "--overlap" in sys.argv
    argv = [a for a in sys.argv if a != "--overlap"]
    with open(argv[1]) as f:
        head = f.readline()
    if head.startswith("index,name,stream,start_us"):  # (kernel names hold commas: split around them)
        rows = []
        for line in open(argv[1]).read().splitlines()[1:]:
            i, rest = line.split(",", 1)
            nm, st, t0, dur = rest.rsplit(",", 3)
            rows.append({"name": nm, "stream": st, "start_us": t0, "dur_us": dur})
    else:
        rows = list(csv.DictReader(open(argv[1])))
    back = int(argv[2]) if len(argv) > 2 else 2
    if rows and "start_us" in rows[0]:
        # the library's own launch events (lab build, FOSVOS_PROF_TIMELINE=<file> around fosvos_profile_start/stop): no
        # profiler on the host, library kernels only
        streams = sorted({r["stream"] for r in rows})
        ks = [dict(q=streams.index(r["stream"]), n=r["name"], s=int(float(r["start_us"]) * 1e3),
                   e=int((float(r["start_us"]) + float(r["dur_us"])) * 1e3), g=0) for r in rows]
    else:
        ks = [dict(q=int(r["Queue_Id"]), n=short(r["Kernel_Name"]), s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]),
                   g=int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows]
    ks.sort(key=lambda k: k["s"])
    sgd = [i for i, k in enumerate(ks) if k["n"].startswith("k_sgd")]
    a, b = sgd[-back - 1], sgd[-back]
    t0 = ks[a]["s"]
    print(f"cycle: {(ks[b]['s'] - t0) / 1e3:.1f} us between optimizer steps, {b - a} kernels")
    if overlap:
        return overlap_report(ks, ks[a]["s"], ks[b]["s"])
    last_end = {}
    for k in ks[a:b + 1]:
        gap = (k["s"] - last_end.get(k["q"], k["s"])) / 1e3
        print(f"q{k['q']} {(k['s'] - t0) / 1e3:9.1f} +{(k['e'] - k['s']) / 1e3:7.1f} us  gap {gap:6.1f}  wg {k['g']:6d}  {k['n']}")
        last_end[k["q"]] = k["e"]


if __name__ == "__main__":
    main()
