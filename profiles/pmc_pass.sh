#!/bin/bash
# Collect PMC counters for bench.py in separate passes (one rocprofv3 run per counter set;
# --pmc is never combined with tracing domains other than --kernel-trace).
# usage: profiles/pmc_pass.sh <out_dir> <bench args...>
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done
echo done
