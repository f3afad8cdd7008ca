set -e
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02o}; mkdir -p $O
for lib in ${2:-libndt_hip_outline512.so} ${3:-libndt_hip_outline768.so}; do
  echo == $lib
  NDT_HIP_LIB=/root/repo/ndt_amd/$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_pipeline or frames_in_sequence or node_pool_overflow" > $O/tests_$lib.log 2>&1 || { tail -5 $O/tests_$lib.log; exit 1; }
  tail -1 $O/tests_$lib.log
  NDT_HIP_LIB=/root/repo/ndt_amd/$lib NDT_HIP_PIPELINE=stream timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu > $O/size_$lib.txt; cat $O/size_$lib.txt
  if grep -q "Memory access fault" $O/*.log $O/*.txt; then exit 1; fi
done
