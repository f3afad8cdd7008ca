#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03s}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "leaf_history or known_answers or item_boxes or (full_resolution and hypercube)" > $O/tests_c5.log 2>&1; tail -2 $O/tests_c5.log
grep -q "failed\|error" $O/tests_c5.log && exit 1
for w in hypercube6d hypercube7d hypercube8d; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}.log 2>&1 || { tail -5 $O/bench_${w}.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}.log") if l.startswith("{")][0])
print("$w", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], flush=True)
PY
done
