#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel name from a counter_collection CSV: pmc_fetch_by_kernel.py <dir> [counter].
FETCH_SIZE is in KiB and, on gfx950, half of the bytes of wide coalesced reads (MI355X_MICROARCH.md): doubled here."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
counter = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
scale = 2.0 * 1024 if counter == "FETCH_SIZE" else (1024.0 if counter == "WRITE_SIZE" else 1.0)
for n, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:70]:70s} launches {c:5d}  {counter} per launch {v * scale / c / 1e6:10.2f} M{'B' if scale != 1 else ''}")
