#!/bin/bash
# Registers, spills and scratch of every kernel of libndt_hip.so, per dimension N = 3 .. 12 (hipcc -Rpass-analysis=kernel-resource-usage
# on the tracked sources; CPU only, no GPU needed).  usage: bash profiles/kernel_resources.sh > profiles/r04_kernel_resources.txt
cd "$(dirname "$0")/../ndt_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage"
tmp=$(mktemp -d)
for n in 3 4 5 6 7 8 9 10 11 12; do
    ( /opt/rocm/bin/hipcc $FLAGS -DNDT_DIMS=$n -c ndt_kernels.hip -o $tmp/k$n.o > $tmp/k$n.log 2>&1 ) &
    if (( n % 4 == 2 )); then wait; fi
done
wait
echo "kernel resources per dimension (gfx950, hipcc $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version.*'))"
echo "columns: VGPRs / VGPR spills / scratch bytes per lane / waves per SIMD"
for n in 3 4 5 6 7 8 9 10 11 12; do
    echo "== N = $n"
    python3 - $tmp/k$n.log <<'PY'
import re, sys
cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    for key in ("VGPRs", "VGPRs Spill", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]"):
        m = re.search(r"remark:\s+%s: (\d+)" % key, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
import subprocess
for name, r in rows.items():
    if not r or "k_" not in name:
        continue
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.split("(")[0].replace("void ", "")
    except Exception:
        dem = name
    print("  %-58s %4d / %4d / %5d / %d" % (dem[-58:], r.get("VGPRs", 0), r.get("VGPRs Spill", 0), r.get("ScratchSize \\[bytes/lane\\]", 0), r.get("Occupancy \\[waves/SIMD\\]", 0)))
PY
done
rm -rf $tmp
