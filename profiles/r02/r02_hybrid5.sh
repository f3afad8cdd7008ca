set -e
cd /root/repo; export TMPDIR=/tmp
for hl in 1 2; do
  echo "== hybrid level $hl"
  NDT_HIP_PIPELINE=hybrid NDT_HIP_HYBRID_LEVEL=$hl NDT_HIP_DEBUG_LEVELS=1 timeout -k 10 200 python profiles/stream_probe.py random4d 1920x1080 2>&1 | grep -v amdgpu | cut -c1-700
done
