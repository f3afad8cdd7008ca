cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02p}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for w in random4d balls4d hypercube3d hypercube6d hypercube8d; do
  for r in 0 16 24 32 40; do
    NDT_HIP_TRACE_REFILL=$r timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}_r$r.log 2>&1 || { tail -3 $O/bench_${w}_r$r.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_r$r.log") if l.startswith("{")][0])
print("$w refill=$r", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"])
PY
  done
done
NDT_HIP_TRACE_REFILL=24 timeout -k 10 200 python profiles/size_probe.py > $O/size_probe_r24.txt 2>&1; cat $O/size_probe_r24.txt
NDT_HIP_PIPELINE=stream timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or oracle or known or full_res or shards or contexts" > $O/tests_stream.log 2>&1; tail -3 $O/tests_stream.log
