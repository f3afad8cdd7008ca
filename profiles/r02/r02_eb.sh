set -e
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02eb}; mkdir -p $O
for w in random4d hypercube3d balls4d; do
  for lib in libndt_hip_eb512.so libndt_hip_eb1024.so libndt_hip.so; do
    NDT_HIP_LIB=/root/repo/ndt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 > $O/bench_${w}_$lib.log 2>&1 || { tail -3 $O/bench_${w}_$lib.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$lib.log") if l.startswith("{")][0])
print("$w $lib", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"])
PY
  done
done
