set -e
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02fu}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -12 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for a in 1 0; do echo "== stream_fused=$a"; NDT_HIP_STREAM_FUSED=$a NDT_HIP_PIPELINE=stream timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu | tee $O/size_$a.txt; done
for a in 1 0; do echo "== stream_fused=$a"; NDT_HIP_STREAM_FUSED=$a timeout -k 10 200 python profiles/shard_probe.py 8 2>&1 | grep -v amdgpu | tail -2; done
