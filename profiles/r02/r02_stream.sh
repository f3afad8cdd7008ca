# the streaming frame kernel on hardware: parity first, then the numbers next to the bounce-synchronous pipeline
set -u
export TMPDIR=/tmp
cd /root/repo 2>/dev/null || true
O=gpurun_out/${1:-r02b}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
rc=$?
tail -25 $O/tests.log
[ $rc -ne 0 ] && exit 1
for w in random4d balls4d hypercube3d hypercube6d hypercube8d; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_stream_$w.log 2>&1 || { tail -5 $O/bench_stream_$w.log; exit 1; }
  NDT_HIP_PIPELINE=levels timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 > $O/bench_levels_$w.log 2>&1 || exit 1
  python - <<PY
import json
for k in ("stream", "levels"):
    d = json.loads([l for l in open("$O/bench_%s_$w.log" % k) if l.startswith("{")][0])
    print("$w", k, "ms/step %.3f" % d["ms_per_step"], "kernel %.3f ms" % d["roofline"]["avg_launch_ms"], "x%g" % d["roofline"]["launches_per_step"], "host_rgba8 %.3f" % d.get("ms_per_step_host_rgba8", 0))
PY
done
timeout -k 10 200 python profiles/size_probe.py > $O/size_probe_stream.txt 2>&1; cat $O/size_probe_stream.txt
