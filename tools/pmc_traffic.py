#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py into profiles/<tag>_pmc_traffic.json.

Units and corrections per /opt/skills/guides (MI355X_MICROARCH.md §HBM, cdna_hip_programming.md §7):
FETCH_SIZE / WRITE_SIZE count KiB at the L2's memory side; on gfx950 FETCH_SIZE reports exactly half of the bytes
of wide (16 B/lane) coalesced streams, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <steps_profiled>"""
import collections
import csv
import glob
import json
import sys

fdir, wdir, out, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])


def load(d, counter):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
    return agg


fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
rows = {}
for n in sorted(set(fetch) | set(write)):
    calls = fetch.get(n, write.get(n))[0]
    fb = 2.0 * fetch.get(n, [0, 0.0])[1] * 1024.0   # gfx950: FETCH_SIZE = 1/2 of a wide coalesced read stream
    wb = write.get(n, [0, 0.0])[1] * 1024.0
    rows[n] = {"launches": calls, "launches_per_step": calls / steps, "hbm_read_bytes_per_launch": fb / calls,
               "hbm_write_bytes_per_launch": wb / calls, "hbm_bytes_per_step": (fb + wb) / steps}
fam = {"igemm": [k for k in rows if "k_conv3x3_igemm" in k],
       "pp": [k for k in rows if "k_conv3x3_pp" in k],  # the persistent forward kernel (round 4)
       "wgrad": [k for k in rows if "k_wgrad3x3" in k or "k_wgrad_first" in k],
       "wgrad_reduce": [k for k in rows if "k_wgrad_fold" in k or "k_wgrad_reduce" in k]}
summary = {}
for name, ks in fam.items():
    launches = sum(rows[k]["launches"] for k in ks)
    tot = sum(rows[k]["hbm_bytes_per_step"] for k in ks) * steps
    summary[name] = {"launches_per_step": launches / steps, "hbm_bytes_per_launch": tot / max(launches, 1),
                     "hbm_bytes_per_step": tot / steps}
conv = ["igemm", "pp", "wgrad"]
summary["conv_mfma_family"] = {
    "launches_per_step": sum(summary[k]["launches_per_step"] for k in conv),
    "hbm_bytes_per_step": sum(summary[k]["hbm_bytes_per_step"] for k in conv),
}
summary["conv_mfma_family"]["hbm_bytes_per_launch"] = (summary["conv_mfma_family"]["hbm_bytes_per_step"] /
                                                       summary["conv_mfma_family"]["launches_per_step"])
summary["all_kernels_hbm_bytes_per_step"] = sum(r["hbm_bytes_per_step"] for r in rows.values())
json.dump({"source": "FOSVOS_BENCH_PRECONDITION=0 rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, tools/round_profiles.sh) -- python3 bench.py "
                     f"--steps {steps - 5} --warmup 5 --no-cpu-baseline --no-roofline --no-infer",
           "corrections": "bytes = KiB * 1024; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B)",
           "steps_profiled": steps, "summary": summary, "per_kernel": rows}, open(out, "w"), indent=1)
for k, v in summary.items():
    print(k, v)
