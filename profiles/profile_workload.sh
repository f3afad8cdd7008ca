#!/bin/bash
# One profiling session of `bench.py --workload W` on the GPU box: kernel times (rocprofv3 --kernel-trace --stats) and the PMC
# passes (each its own run; --pmc is never combined with other trace domains), then profiles/r04_profile_<W>_<H>p.json
# (what bench.py's roofline object reads; it carries the hash of the library it was measured on) and the summaries
# under profiles/.   usage: bash profiles/profile_workload.sh <workload> [bench args...]
set -u
export TMPDIR=/tmp
cd /root/repo 2>/dev/null || true
w=$1; shift
O=gpurun_out/prof_$w
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 "$@" > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
bash profiles/pmc_pass.sh $O/pmc1 --workload $w --steps 4 --warmup 1 "$@" || exit 1
python3 profiles/make_profile_json.py $O profiles/r04_profile_${w}_1080p.json 2073600 > $O/profile.json.log || exit 1
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) profiles/r04_bench_${w}_1080p_kernel_stats.csv
python3 profiles/pmc_summary.py $O/pmc1 > profiles/r04_pmc_${w}_1080p.txt 2>&1
# the bench line once more, now that the profile of this very library exists: its roofline object carries the measured traffic
timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 "$@" 2>/dev/null | grep '^{' > profiles/r04_bench_${w}_1080p.json.log || exit 1
python3 - <<PY
import json
d = json.load(open("profiles/r04_profile_${w}_1080p.json"))
print("$w:", d["kernel"][:50], "avg %.1f us, hbm %.1f MB/launch, fp64 issue %.2f, lane util %.2f, wait %.2f" % (
    d["avg_launch_ns"] / 1e3, d.get("hbm_bytes_per_launch", 0) / 1e6, d.get("fp64_valu_issue_frac", 0), d.get("lane_utilisation", 0), d.get("wait_frac_of_wave_cycles", 0)))
PY
