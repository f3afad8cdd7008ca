cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02v}; mkdir -p $O
NDT_HIP_LIB=/root/repo/ndt_amd/libndt_hip_timing.so timeout -k 10 200 python profiles/levels_probe.py > $O/phase_c3.txt 2>&1; grep "ndt_hip" $O/phase_c3.txt | cut -c1-400
NDT_HIP_LIB=/root/repo/ndt_amd/libndt_hip_timing.so timeout -k 10 200 python profiles/levels_probe.py --res 64x36 > $O/phase_c3_tiny.txt 2>&1; grep "ndt_hip" $O/phase_c3_tiny.txt | cut -c1-400
