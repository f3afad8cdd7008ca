cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02e}; mkdir -p $O
run() { # name lib env...
  n=$1; shift; lib=$1; shift
  env NDT_HIP_LIB=/root/repo/ndt_amd/$lib "$@" timeout -k 10 120 python profiles/stream_probe.py ${W:-random4d} ${S:-1920x1080} > $O/probe_$n.txt 2>&1; echo "--- $n"; grep -v amdgpu.ids $O/probe_$n.txt | sed 's/wavefronts by the 32nd.*//' | cut -c1-900
}
run c3_768 libndt_hip_knobs768.so
run c3_512 libndt_hip_knobs.so
W=balls4d run balls_768 libndt_hip_knobs768.so
W=balls4d run balls_512 libndt_hip_knobs.so
W=hypercube3d run h3_768 libndt_hip_knobs768.so
W=hypercube3d run h3_512 libndt_hip_knobs.so
S=64x36 run c3tiny libndt_hip_knobs.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
