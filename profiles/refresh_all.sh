#!/bin/bash
# Everything the numbers in DESIGN.md / README.md come from, on one GPU box: the GPU tests, bench.py for every workload,
# the rocprofv3 sessions (kernel stats + PMC passes -> profiles/r04_profile_*.json, which bench.py's roofline object
# quotes when the library hash matches), the frame-time-vs-size sweep for both pipelines and the strong-scaling shard probe.
#   gpurun --timeout 1200 -- 'PART=1 bash profiles/refresh_all.sh'; gpurun --timeout 1200 -- 'PART=2 bash profiles/refresh_all.sh'      then copy gpurun_out/r04_final/* of interest into profiles/
set -u
export TMPDIR=/tmp
cd /root/repo 2>/dev/null || true
O=gpurun_out/r04_final
mkdir -p $O
# only what THIS call wrote goes back (the other part's files in the snapshot may be older than what that part last produced)
touch $O/.start
give_back() { find profiles -maxdepth 1 -name 'r04_*' -newer $O/.start -exec cp {} $O/ \; 2>/dev/null; }
trap give_back EXIT
# in two calls (gpurun's limit is 20 minutes): PART=1 the GPU tests + the three 3-D / 4-D workloads, PART=2 the 6-D .. 8-D sweep + the probes
PART=${PART:-1}
if [ "$PART" = 1 ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
grep "bytes differ\|noisy values\|per-value t\|reference-equivalent rays\|sampler:\|inside their runs" $O/tests.log > profiles/r04_gpu_test_notes.txt; tail -2 $O/tests.log >> profiles/r04_gpu_test_notes.txt; tail -2 $O/tests.log
for w in random4d balls4d hypercube3d; do
  bash profiles/profile_workload.sh $w || exit 1
done
give_back
exit 0
fi
if [ "$PART" = 3 ]; then
# the zoo by dimension and by face cull, the exit probe and kernel timeline of the benchmark frame, the soak
timeout -k 10 300 python profiles/zoo_cull_probe.py 2>&1 | grep -v amdgpu > profiles/r04_zoo_by_face_cull.txt; cat profiles/r04_zoo_by_face_cull.txt
timeout -k 10 200 python profiles/zoo_probe.py 2>&1 | grep -v amdgpu > $O/r04_zoo_probe.txt
timeout -k 10 120 python3 profiles/coop_probe.py c3_random4d 1920x1080 1 --probe --off-only 2>&1 | grep -v "amdgpu\|SIMD 0/1" | cut -c1-400 > profiles/r04_exit_probe_c3.txt
timeout -k 10 400 python profiles/soak.py 150 > $O/soak.log 2>&1; tail -12 $O/soak.log > profiles/r04_soak.txt; tail -3 profiles/r04_soak.txt
give_back
exit 0
fi
for w in hypercube6d hypercube7d hypercube8d; do
  bash profiles/profile_workload.sh $w || exit 1
done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/bench_default.log 2>&1 || exit 1
grep '^{' $O/bench_default.log > profiles/r04_bench_default.json.log
for w in balls4d hypercube3d hypercube6d hypercube7d hypercube8d; do
  NDT_HIP_PIPELINE=stream timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 2>/dev/null | grep '^{' > profiles/r04_bench_stream_$w.json.log
done
for pl in levels stream; do
  NDT_HIP_PIPELINE=$pl timeout -k 10 200 python profiles/size_probe.py > $O/size_probe_$pl.txt 2>&1
  grep -v amdgpu $O/size_probe_$pl.txt > profiles/r04_frame_time_vs_size_$pl.txt
done
timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu > profiles/r04_frame_time_vs_size_auto.txt
timeout -k 10 300 python profiles/shard_probe.py 1 2 4 8 2>&1 | grep -v amdgpu > profiles/r04_shard_probe_strong.txt; cat profiles/r04_shard_probe_strong.txt
NDT_HIP_PIPELINE=levels timeout -k 10 200 python profiles/shard_probe.py 8 2>&1 | grep -v amdgpu > profiles/r04_shard_probe_strong_levels_n8.txt
NDT_HIP_STREAM_PROBE=1 timeout -k 10 100 python profiles/stream_probe.py random4d 64x36 960x540 2>&1 | grep -v amdgpu > profiles/r04_stream_probe_random4d.txt
give_back
tail -1 profiles/r04_bench_default.json.log | cut -c1-400
