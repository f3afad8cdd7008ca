#!/bin/bash
# Kernel statistics of one workload for library variants: bash profiles/scripts/r03_kstats.sh <outdir> <workload> lib1.so lib2.so ...
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O; W=$2; shift 2
for lib in "$@"; do
  export NDT_HIP_LIB=/root/repo/ndt_amd/$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ks_${W}_$lib -o run --output-format csv -- python3 bench.py --no-cpu-baseline --workload $W --steps 30 --warmup 3 > $O/ks_${W}_$lib.log 2>&1 || { tail -5 $O/ks_${W}_$lib.log; exit 1; }
  f=$(find $O/ks_${W}_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $W $lib"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("  %-50s calls %5s avg %9.1f us  min %8.1f  max %8.1f  %5.1f %%" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
done
