# round 2, first GPU call: the split library + the thicker goldens on hardware, and the numbers the round starts from
set -u
export TMPDIR=/tmp
cd /root/repo 2>/dev/null || true
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 > $O/bench_c3.log 2>&1 || exit 1
for w in hypercube6d hypercube7d hypercube8d; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 2 > $O/bench_$w.log 2>&1 || exit 1
done
bash profiles/pmc_pass.sh $O/pmc_8d --workload hypercube8d --steps 3 --warmup 1 || exit 1
bash profiles/pmc_pass.sh $O/pmc_6d --workload hypercube6d --steps 3 --warmup 1 || exit 1
python profiles/size_probe.py > $O/size_probe.txt 2>&1
tail -2 $O/bench_c3.log
