#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ counter CSVs per kernel dispatch group (Grid_Size, LDS size): ratios that show
where wave time goes.  usage: pmc_sq.py <counter_collection.csv> [<second pass csv>]"""
import collections
import csv
import sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "k_conv3x3_igemm" not in name and "k_wgrad" not in name and "k_conv3x3_pp" not in name:
            continue
        short = name.split("<", 1)[1].split(">(")[0] if "<" in name else name
        key = (short[:40], int(r["Grid_Size"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    tabs = [load(p) for p in sys.argv[1:]]
    keys = list(tabs[0].keys())
    for k in keys:
        c = {}
        for t in tabs:
            for n, v in t.get(k, {}).items():
                c[n] = sum(v) / len(v)
        print(f"{k[0]:40s} grid {k[1]:8d} lds {k[2]:6d} regs {k[3]:4d} n={len(next(iter(tabs[0][k].values())))}")
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc:
            print("   of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f  wait_inst_lds %.3f" % (
                c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_LDS"] / wc))
            print("   busy_cycles %.3e  wave_cycles %.3e  waves-per-busy-cycle %.2f  mfma_busy/busy %.3f  mfma insts %.3e" % (
                c["SQ_BUSY_CYCLES"], wc, wc / c["SQ_BUSY_CYCLES"], c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], c["SQ_INSTS_MFMA"]))
        if "SQ_LDS_IDX_ACTIVE" in c:
            print("   lds_idx_active %.3e  bank_conflict %.3e  active_inst_lds %.3e insts_lds %.3e  gui_active %.3e  vmem level %.3e active_vmem %.3e vmem_rd %.3e" % (
                c["SQ_LDS_IDX_ACTIVE"], c["SQ_LDS_BANK_CONFLICT"], c["SQ_ACTIVE_INST_LDS"], c["SQ_INSTS_LDS"], c["GRBM_GUI_ACTIVE"],
                c["SQ_INST_LEVEL_VMEM"], c["SQ_ACTIVE_INST_VMEM"], c["SQ_INSTS_VMEM_RD"]))


if __name__ == "__main__":
    main()
