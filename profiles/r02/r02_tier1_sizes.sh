set -e
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02t1}; mkdir -p $O
for sc in c5_hypercube6d c5_hypercube7d c5_hypercube8d; do
  for pl in levels stream; do
    echo "== $sc $pl"
    NDT_HIP_PIPELINE=$pl timeout -k 10 280 python profiles/size_probe.py $sc 128 2>&1 | grep -v amdgpu | tee $O/size_${sc}_$pl.txt
  done
done
