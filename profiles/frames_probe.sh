#!/bin/bash
# Frame-level parallelism of the host driver: ndt_hip -j K on an animation (development aid / DESIGN.md numbers).
# usage: profiles/frames_probe.sh <scene> <dims> <frames> <depth>
scene=${1:-hypercube}; dims=${2:-3}; n=${3:-48}; depth=${4:-128}
so=oracle/_ref/scenes/$scene.so
out=$(mktemp -d)
for j in 1 2 4 6; do
  ( cd $out && timeout -k 10 300 $OLDPWD/ndt_amd/host/ndt_hip -s $OLDPWD/$so -d $dims -r 1920x1080 -l $depth -f 0:$((n-1)) -j $j 2>&1 | grep "frames in" | sed "s/^/$scene ${dims}-D 1080p -j $j: /" )
done
rm -rf $out
