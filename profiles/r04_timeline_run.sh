#!/bin/bash
# one benchmark frame's kernel timeline (rocprofv3 --kernel-trace) + the exit probe of its trace launches
# usage (GPU box): bash profiles/r04_timeline_run.sh [workload]   -> gpurun_out/r04_timeline_<workload>.txt, gpurun_out/r04_exit_probe_c3.txt
w=${1:-random4d}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --workload $w --no-cpu-baseline > /tmp/tl.log 2>&1 || { tail -5 /tmp/tl.log; exit 1; }
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python3 profiles/timeline.py $f > gpurun_out/r04_timeline_$w.txt
python3 profiles/kernel_medians.py $f >> gpurun_out/r04_timeline_$w.txt
if [ "$w" = random4d ]; then timeout -k 10 120 python3 profiles/coop_probe.py c3_random4d 1920x1080 1 --probe --off-only > gpurun_out/r04_exit_probe_c3.txt 2>&1; fi
cat gpurun_out/r04_timeline_$w.txt
