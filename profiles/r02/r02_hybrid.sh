cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02s}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
NDT_HIP_PIPELINE=hybrid timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or oracle or full_res or shards or contexts or sequence or overflow or depth or anti" > $O/tests_hybrid.log 2>&1; tail -3 $O/tests_hybrid.log
for w in random4d balls4d hypercube3d hypercube6d hypercube8d; do
  for pl in levels hybrid; do
    for H in 1 2; do
      [ $pl = levels ] && [ $H = 2 ] && continue
      NDT_HIP_PIPELINE=$pl NDT_HIP_HYBRID_LEVEL=$H timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 > $O/bench_${w}_${pl}$H.log 2>&1 || { tail -3 $O/bench_${w}_${pl}$H.log; exit 1; }
      python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_${pl}$H.log") if l.startswith("{")][0])
print("$w $pl H=$H", "ms/step %.3f" % d["ms_per_step"], "kernels %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"], "host_rgba8 %.3f" % d.get("ms_per_step_host_rgba8", 0))
PY
    done
  done
done
NDT_HIP_PIPELINE=hybrid timeout -k 10 200 python profiles/size_probe.py > $O/size_probe_hybrid.txt 2>&1; cat $O/size_probe_hybrid.txt
