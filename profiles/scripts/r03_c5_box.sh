#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03g}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "leaf_history or known_answers or (full_resolution and hypercube)" > $O/tests_c5.log 2>&1; tail -3 $O/tests_c5.log
grep -q "failed\|error" $O/tests_c5.log && exit 1
for w in hypercube6d hypercube7d hypercube8d; do
  for v in "0 1 64" "1 1 64" "1 1 48"; do
    set -- $v
    NDT_HIP_LEAF_SCAN=$1 NDT_HIP_ITEM_BOXES=$2 NDT_HIP_LEAF_SCAN_GROUP=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}_$1_$2_$3.log 2>&1 || { tail -5 $O/bench_${w}_$1_$2_$3.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$1_$2_$3.log") if l.startswith("{")][0])
print("$w leaf_scan=$1 item_boxes=$2 group>=$3", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"], flush=True)
PY
  done
done
