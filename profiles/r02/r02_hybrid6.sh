set -e
cd /root/repo; export TMPDIR=/tmp
NDT_HIP_PIPELINE=stream timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu
for hl in 1 2; do
  echo "== hybrid level $hl"
  NDT_HIP_PIPELINE=hybrid NDT_HIP_HYBRID_LEVEL=$hl timeout -k 10 200 python profiles/size_probe.py 2>&1 | grep -v amdgpu | tail -3
done
