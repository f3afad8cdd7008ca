cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02r}; mkdir -p $O
NDT_HIP_PIPELINE=levels timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hypercube or zoo6d" > $O/tests_mid.log 2>&1; tail -3 $O/tests_mid.log
for w in hypercube6d hypercube7d hypercube8d; do
  for mid in 0 1; do
    if [ $mid = 0 ]; then export NDT_HIP_NO_MID_TIER=1; else unset NDT_HIP_NO_MID_TIER; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}_mid$mid.log 2>&1 || { tail -3 $O/bench_${w}_mid$mid.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_mid$mid.log") if l.startswith("{")][0])
print("$w mid=$mid", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"])
PY
  done
done
