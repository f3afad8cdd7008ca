cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02l}; mkdir -p $O
for w in random4d hypercube3d; do
NDT_HIP_PIPELINE=levels timeout -k 10 200 python profiles/multi_ctx_probe.py $w 1920x1080 1 2 3 4 6 8 2>&1 | grep -v amdgpu | tee -a $O/multi.txt
done
NDT_HIP_PIPELINE=levels timeout -k 10 200 python profiles/multi_ctx_probe.py random4d 3840x2160 1 2 4 8 2>&1 | grep -v amdgpu | tee -a $O/multi.txt
timeout -k 10 200 python profiles/multi_ctx_probe.py random4d 1920x1080 1 2 4 2>&1 | grep -v amdgpu | tee -a $O/multi.txt
