#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03e}; mkdir -p $O
for sc in c5_hypercube8d c5_hypercube6d; do
NDT_HIP_LIB=/root/repo/ndt_amd/libndt_hip_timing.so timeout -k 10 200 python profiles/levels_probe.py --scene $sc --depth 128 > $O/phase_${sc}.txt 2>&1; echo == $sc; grep "ndt_hip" $O/phase_${sc}.txt | grep -v "bounce\|shade_emit" | cut -c1-330
done
