#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "leaf_history or item_boxes or every_pipeline or framebuffer_vs" 2>&1 | tail -2
for pl in levels stream; do for sc in c5_hypercube6d c5_hypercube8d; do echo "== $sc $pl"; NDT_HIP_PIPELINE=$pl timeout -k 10 200 python profiles/size_probe.py $sc 128 2>&1 | grep -v amdgpu | head -5; done; done
for w in random4d hypercube3d balls4d; do timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w ms/step %.3f' % d['ms_per_step'])"; done
