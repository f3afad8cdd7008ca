cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02t}; mkdir -p $O
NDT_HIP_DEBUG_LEVELS=1 NDT_HIP_PIPELINE=hybrid timeout -k 10 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden and zoo3d_mirror" > $O/t_mirror.log 2>&1; grep "overflow\|passed\|failed" $O/t_mirror.log | head -12
for w in random4d balls4d hypercube3d; do
  for pl in levels hybrid stream; do
      NDT_HIP_PIPELINE=$pl timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 > $O/bench_${w}_${pl}.log 2>&1 || { tail -3 $O/bench_${w}_${pl}.log; exit 1; }
      python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_${pl}.log") if l.startswith("{")][0])
print("$w $pl", "ms/step %.3f" % d["ms_per_step"], "kernels %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"])
PY
  done
done
NDT_HIP_PIPELINE=stream timeout -k 10 200 python profiles/size_probe.py > $O/size_probe_stream.txt 2>&1; cat $O/size_probe_stream.txt
