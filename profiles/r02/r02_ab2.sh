# A/B of two builds of the library: bash profiles/r02_ab2.sh <outdir> <libA> <libB>
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02ab}; mkdir -p $O
A=${2:-libndt_hip.so}; B=${3:-libndt_hip_nopeephole.so}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in random4d hypercube3d balls4d hypercube6d hypercube8d; do
  for lib in $A $B; do
    NDT_HIP_LIB=/root/repo/ndt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 > $O/bench_${w}_$lib.log 2>&1 || { tail -3 $O/bench_${w}_$lib.log; exit 1; }
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${w}_$lib.log") if l.startswith("{")][0])
print("$w $lib", "ms/step %.3f" % d["ms_per_step"], "trace %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"])
PY
  done
done
for lib in $A $B; do NDT_HIP_LIB=/root/repo/ndt_amd/$lib timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu > $O/size_$lib.txt; cat $O/size_$lib.txt; done
