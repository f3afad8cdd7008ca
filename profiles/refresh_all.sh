set -u
export TMPDIR=/tmp
cd /root/repo 2>/dev/null || true
mkdir -p gpurun_out/r01v8
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01v8/stats -- python3 bench.py --steps 20 --warmup 3 > gpurun_out/r01v8/bench_c3.log 2>&1 || exit 1
bash profiles/pmc_pass.sh gpurun_out/r01v8/pmc --steps 4 --warmup 1 || exit 1
bash profiles/pmc_pass2.sh gpurun_out/r01v8/pmc2 --steps 4 --warmup 1 || exit 1
timeout -k 10 200 python bench.py --workload balls4d --steps 20 --warmup 3 > gpurun_out/r01v8/bench_c2.log 2>&1
timeout -k 10 200 python bench.py --workload hypercube3d --steps 20 --warmup 3 > gpurun_out/r01v8/bench_c1.log 2>&1
find gpurun_out/r01v8 -name "*.csv" | head -30
bash profiles/pmc_pass3.sh gpurun_out/r01v8/pmc3 --steps 4 --warmup 1
python3 profiles/timeline.py $(ls gpurun_out/r01v8/stats/*/*kernel_trace.csv | tail -1) > gpurun_out/r01v8/timeline.txt
