#!/bin/bash
# the whole GPU suite, then the bench line of every workload (no rocprof)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03k}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -q "failed\|error" $O/tests.log && exit 1
for w in random4d balls4d hypercube3d hypercube6d hypercube7d hypercube8d; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}.log 2>&1 || { tail -5 $O/bench_${w}.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}.log") if l.startswith("{")][0])
print("$w", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"], "frac %.3f" % d["roofline"]["frac"], flush=True)
PY
done
