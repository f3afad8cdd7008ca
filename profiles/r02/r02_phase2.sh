cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02p2}; mkdir -p $O
for sc in c5_hypercube6d c5_hypercube8d c2_balls4d; do
NDT_HIP_LIB=/root/repo/ndt_amd/libndt_hip_timing.so timeout -k 10 200 python profiles/levels_probe.py --scene $sc > $O/phase_$sc.txt 2>&1; echo == $sc; grep "ndt_hip" $O/phase_$sc.txt | grep -v "bounce\|trace launch\|shade_emit" | cut -c1-330
done
