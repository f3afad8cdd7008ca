cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02u}; mkdir -p $O
NDT_HIP_PIPELINE=hybrid NDT_HIP_DEBUG_LEVELS=1 timeout -k 10 120 python profiles/stream_probe.py random4d 1920x1080 > $O/probe_hybrid.txt 2>&1; grep -v amdgpu $O/probe_hybrid.txt | sed 's/wavefronts by the 32nd.*//' | cut -c1-900
NDT_HIP_PIPELINE=hybrid timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or oracle or full_res or shards or contexts or sequence or overflow or depth or anti" > $O/tests_hybrid.log 2>&1; tail -3 $O/tests_hybrid.log
