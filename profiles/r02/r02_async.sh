set -e
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02as}; mkdir -p $O
NDT_HIP_FINISH_ASYNC=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -8 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for w in random4d hypercube3d balls4d hypercube6d hypercube8d; do
  for a in 1 0; do
    NDT_HIP_FINISH_ASYNC=$a timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 > $O/bench_${w}_$a.log 2>&1 || { tail -3 $O/bench_${w}_$a.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$a.log") if l.startswith("{")][0])
print("$w finish_async=$a", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"])
PY
  done
done
for a in 1 0; do NDT_HIP_PIPELINE=levels NDT_HIP_FINISH_ASYNC=$a timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu > $O/size_$a.txt; cat $O/size_$a.txt; done
