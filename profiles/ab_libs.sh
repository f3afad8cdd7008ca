#!/bin/bash
# usage: bash profiles/ab_libs.sh tag lib1 lib2 ...   (libs by suffix: ndt_amd/libndt_hip_<suffix>.so; "new" = the default library)
tag=$1; shift
for v in "$@"; do
    if [ "$v" = new ]; then NDT_HIP_DEBUG_LEVELS=1 timeout -k 10 150 python profiles/ab_probe.py > gpurun_out/${tag}_ab_$v.log 2>&1 || exit 1
    else NDT_HIP_LIB=$PWD/ndt_amd/libndt_hip_$v.so timeout -k 10 150 python profiles/ab_probe.py > gpurun_out/${tag}_ab_$v.log 2>&1 || exit 1; fi
done
first=$1
grep -v "amdgpu\|library\|ndt_hip:" gpurun_out/${tag}_ab_$first.log | cut -c1-31 > /tmp/ab_cols.txt
for v in "$@"; do grep -v "amdgpu\|library\|ndt_hip:" gpurun_out/${tag}_ab_$v.log | cut -c33-43 | paste -d"|" /tmp/ab_cols.txt - > /tmp/ab_cols2.txt; mv /tmp/ab_cols2.txt /tmp/ab_cols.txt; done
echo "columns: $*"; cat /tmp/ab_cols.txt
grep -h "scene blob" gpurun_out/${tag}_ab_new.log 2>/dev/null | sort -u
