#!/usr/bin/env python3
"""Turn one profiling session of `bench.py --workload W` (profiles/profile_workload.sh: a rocprofv3 --kernel-trace --stats
run + the PMC passes of profiles/pmc_pass.sh, every pass a separate run) into the JSON bench.py reads for its roofline
object: per launch of the dominant kernel, its average duration, the HBM bytes it moved (FETCH_SIZE / WRITE_SIZE, in KB;
on gfx950 FETCH_SIZE tallies a 128-byte fabric read as 64 bytes: x2, checked on k_finish_pixels whose bytes are known),
and the secondary bound SURVEY 8(d) asks for -- FP64 vector issue: SQ_INSTS_VALU x 4 cycles (a 64-lane FP64 instruction
occupies its SIMD for four) / (SIMDs x launch time x clock) -- with lane utilisation and the share of wave-cycles spent waiting.

usage: make_profile_json.py <session dir> <out.json> <pixels> [dominant kernel substring]"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, out, pixels = sys.argv[1], sys.argv[2], float(sys.argv[3])
want = sys.argv[4] if len(sys.argv) > 4 else None
SIMDS, CLOCK_HZ = 1024, 2.4e9          # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz

# kernel time: the --stats run
stats = {}
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            stats[row["Name"]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "pct": float(row["Percentage"])}
if not stats:
    raise SystemExit("no kernel_stats.csv under %s/stats" % d)
if want:
    dom = max((k for k in stats if want in k), key=lambda k: stats[k]["pct"])
else:
    dom = max(stats, key=lambda k: stats[k]["pct"])
tot = defaultdict(lambda: defaultdict(float))
n = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "pmc*", "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = "dom" if row["Kernel_Name"] == dom else ("finish" if "k_finish_pixels" in row["Kernel_Name"] else None)
            if k:
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                n[k][row["Counter_Name"]] += 1


def per_launch(k, name):
    return tot[k][name] / n[k][name] if n[k].get(name) else None


res = {"kernel": dom, "calls": stats[dom]["calls"], "avg_launch_ns": stats[dom]["avg_ns"], "share_of_gpu_time_pct": stats[dom]["pct"],
       "kernels": {k.split("(")[0][-48:]: v for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["pct"])[:8]}}
f_raw, w_raw = per_launch("dom", "FETCH_SIZE"), per_launch("dom", "WRITE_SIZE")
if f_raw is not None and w_raw is not None:
    res["fetch_size_bytes_raw"] = f_raw * 1024.0
    res["write_size_bytes"] = w_raw * 1024.0
    res["hbm_bytes_per_launch"] = 2.0 * f_raw * 1024.0 + w_raw * 1024.0
    ff, fw = per_launch("finish", "FETCH_SIZE"), per_launch("finish", "WRITE_SIZE")
    if ff and fw:
        # k_finish_pixels reads 36 and writes 32 bytes per pixel: the check of the x2
        res["fetch_correction_measured_on_finish_pixels"] = 36.0 * pixels / (ff * 1024.0)
        res["write_ratio_measured_on_finish_pixels"] = fw * 1024.0 / (32.0 * pixels)
valu, act, thr = per_launch("dom", "SQ_INSTS_VALU"), per_launch("dom", "SQ_ACTIVE_INST_VALU"), per_launch("dom", "SQ_THREAD_CYCLES_VALU")
if valu:
    t = stats[dom]["avg_ns"] * 1e-9
    res["valu_insts_per_launch"] = valu
    res["fp64_valu_issue_frac"] = valu * 4.0 / (SIMDS * t * CLOCK_HZ)
    if act and thr:
        res["lane_utilisation"] = thr / (act * 64.0)
    salu = per_launch("dom", "SQ_INSTS_SALU")
    if salu:
        res["salu_per_valu"] = salu / valu
    wait, cyc = per_launch("dom", "SQ_WAIT_ANY"), per_launch("dom", "SQ_WAVE_CYCLES")
    if wait and cyc:
        res["wait_frac_of_wave_cycles"] = wait / cyc
lib = os.path.join(ROOT, "ndt_amd", "libndt_hip.so")
res["lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
with open(out, "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
    fh.write("\n")
print(json.dumps(res, indent=1, sort_keys=True))
