#!/bin/bash
# End to end with the host program (scene program on the CPU every frame -> flatten -> upload -> render): an animation of
# scenes/hypercube.c, frames in flight -j 1 / 2 / 4.  bash profiles/scripts/r03_anim.sh [outdir]
cd /root/repo; O=gpurun_out/${1:-r03_anim}; mkdir -p $O
for d in 4 6 8; do for j in 1 2 4; do
  t0=$(date +%s.%N)
  timeout -k 10 300 ./ndt_amd/host/ndt_hip -s oracle/_ref/scenes/hypercube.so -d $d -r 1080p -f 0:23:300 -l 8 -j $j -t ${T:-1} > $O/anim_${d}_$j.log 2>&1; rc=$?
  t1=$(date +%s.%N)
  echo "hypercube $d-D, 24 frames of 1920x1080, -j $j -t ${T:-1}: rc $rc, $(echo "$t1 $t0" | awk '{printf "%.2f s wall, %.1f ms a frame", $1-$2, ($1-$2)*1000/24}')"
done; done
tail -4 $O/anim_8_1.log | cut -c1-250
