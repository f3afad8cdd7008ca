set -e
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02t1b}; mkdir -p $O
A=${2:-libndt_hip.so}; B=${3:-libndt_hip_base.so}
NDT_HIP_LIB=/root/repo/ndt_amd/$A timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_hull_box.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for w in balls4d hypercube6d hypercube7d hypercube8d; do
  for lib in $A $B; do
    NDT_HIP_LIB=/root/repo/ndt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_${w}_$lib.log 2>&1 || { tail -3 $O/bench_${w}_$lib.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$lib.log") if l.startswith("{")][0])
print("$w $lib", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"])
PY
  done
done
