#!/bin/bash
# The modes beyond the deterministic frame at full size: recursive AA, -n samples, stereo, VR, depth map.
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r03_modes}; mkdir -p $O
for sc in c3_random4d c2_balls4d; do
  timeout -k 10 200 python profiles/aa_probe.py $sc 1920x1080 20,4 4 2>&1 | grep -v amdgpu | tee -a $O/aa.txt
done
timeout -k 10 300 python profiles/samples_probe.py ns_zoo4d_dof 2>&1 | grep -v amdgpu | tee $O/samples.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/aa_prof -o run --output-format csv -- python3 profiles/aa_probe.py c3_random4d 1920x1080 20,4 4 > $O/aa_prof.log 2>&1
f=$(find $O/aa_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 $f | cut -c1-200 | tee $O/aa_kernel_stats.txt
