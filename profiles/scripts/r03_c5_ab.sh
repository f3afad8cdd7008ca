#!/bin/bash
# round 3, C5 A/B: leaf history (VisitMask<0>) against the slab, per-bounce pipeline; then the GPU tests.
#   gpurun --timeout 900 -- 'bash profiles/scripts/r03_c5_ab.sh r03a'
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03a}; mkdir -p $O
for w in hypercube6d hypercube7d hypercube8d; do
  for h in 4 0; do
    NDT_HIP_LEAF_HISTORY=$h timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}_h$h.log 2>&1 || { tail -5 $O/bench_${w}_h$h.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_h$h.log") if l.startswith("{")][0])
print("$w leaf_history=$h", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"], flush=True)
PY
  done
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; tail -3 $O/tests.log
