#!/bin/bash
# third counter family: vector L1 / L2 behaviour (is the traversal stack in scratch served by L2?).
# usage: profiles/pmc_pass3.sh <out_dir> <bench args...>
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done
echo done
