#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03r}; mkdir -p $O
for w in hypercube6d hypercube8d; do
  for g in 64 48 32 24 16 8; do
    NDT_HIP_LEAF_SCAN_GROUP=$g timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}_$g.log 2>&1 || { tail -5 $O/bench_${w}_$g.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$g.log") if l.startswith("{")][0])
print("$w group>=$g", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], flush=True)
PY
  done
done
