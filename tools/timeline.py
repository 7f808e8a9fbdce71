#!/usr/bin/env python3
"""Timeline of one accumulation cycle from a rocprofv3 --kernel-trace CSV: per HIP queue, the kernels in start order with
gaps, between two optimizer steps (k_sgd).   usage: timeline.py trace.csv [cycle-index-from-end]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ks = [dict(q=int(r["Queue_Id"]), n=short(r["Kernel_Name"]), s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]),
               g=int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows]
    ks.sort(key=lambda k: k["s"])
    sgd = [i for i, k in enumerate(ks) if k["n"].startswith("k_sgd")]
    a, b = sgd[-back - 1], sgd[-back]
    t0 = ks[a]["s"]
    print(f"cycle: {(ks[b]['s'] - t0) / 1e3:.1f} us between optimizer steps, {b - a} kernels")
    last_end = {}
    for k in ks[a:b + 1]:
        gap = (k["s"] - last_end.get(k["q"], k["s"])) / 1e3
        print(f"q{k['q']} {(k['s'] - t0) / 1e3:9.1f} +{(k['e'] - k['s']) / 1e3:7.1f} us  gap {gap:6.1f}  wg {k['g']:6d}  {k['n']}")
        last_end[k["q"]] = k["e"]


if __name__ == "__main__":
    main()
