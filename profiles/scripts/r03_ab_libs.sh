#!/bin/bash
# A/B of library variants: bash profiles/scripts/r03_ab_libs.sh <outdir> "<workloads>" lib1.so lib2.so ...
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O; W="$2"; shift 2
for w in $W; do
  for lib in "$@"; do
    NDT_HIP_LIB=/root/repo/ndt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}_$lib.log 2>&1 || { tail -5 $O/bench_${w}_$lib.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$lib.log") if l.startswith("{")][0])
print("$w $lib", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"], flush=True)
PY
  done
done
