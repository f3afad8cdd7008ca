#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03p}; mkdir -p $O
NDT_HIP_GATE_PREPASS=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "known_answers or framebuffer_vs_reference or item_sets or (full_resolution and not hypercube)" > $O/tests_pp.log 2>&1; tail -2 $O/tests_pp.log
for w in random4d hypercube3d; do
  for v in 0 1; do
    NDT_HIP_GATE_PREPASS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 > $O/bench_${w}_$v.log 2>&1 || { tail -5 $O/bench_${w}_$v.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$v.log") if l.startswith("{")][0])
print("$w prepass=$v", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"], flush=True)
PY
  done
done
for v in 0 1; do echo prepass=$v; NDT_HIP_GATE_PREPASS=$v timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu | head -4; done
