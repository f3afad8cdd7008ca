#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03e}; mkdir -p $O
for sc in c5_hypercube8d; do
for ls in "0 0" "0 1"; do
set -- $ls
NDT_HIP_LEAF_SCAN=$1 NDT_HIP_ITEM_BOXES=$2 NDT_HIP_LIB=/root/repo/ndt_amd/libndt_hip_timing.so timeout -k 10 200 python profiles/levels_probe.py --scene $sc --depth 128 > $O/phase_${sc}_ls$1_ib$2.txt 2>&1; echo == $sc leaf_scan=$1 boxes=$2; grep "ndt_hip" $O/phase_${sc}_ls$1_ib$2.txt | grep -v "bounce\|shade_emit" | cut -c1-330
done
done
