#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (profiles/pmc_pass.sh) per kernel: sum of each counter over
dispatches, dispatch count, and derived ratios.  usage: pmc_summary.py <dir> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if filt and filt not in k:
                continue
            k = k.split("(")[0][-60:]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
    c = tot[k]
    print("==", k)
    for name in sorted(c):
        print("   %-26s %16.0f  (%d dispatches)" % (name, c[name], calls[k][name]))
    if c.get("SQ_ACTIVE_INST_VALU"):
        print("   lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = %.3f" % (
            c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)))
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_INSTS_VALU"):
        print("   VALU insts per wave = %.0f ; wave cycles(quad) per wave = %.0f" % (
            c["SQ_INSTS_VALU"] / max(1, c["SQ_WAVES"]), c["SQ_WAVE_CYCLES"] / max(1, c["SQ_WAVES"])))
