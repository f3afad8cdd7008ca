#!/bin/bash
# second counter family: latencies / occupancy.  usage: profiles/pmc_pass2.sh <out_dir> <bench args...>
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_FLAT" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done
echo done
