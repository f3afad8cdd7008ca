#!/bin/bash
# round 4: the batch cull (ndt_device.hpp:batch_dead_items) -- its tests, then frame times with and without it
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_cull or cooperative or every_pipeline or golden or full_res" > gpurun_out/r04p_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04p_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 200 python profiles/fusion_probe.py --opt=batch_cull > gpurun_out/r04p_cull.log 2>&1; grep -v amdgpu gpurun_out/r04p_cull.log | head -12
timeout -k 10 200 python profiles/fusion_probe.py --opt=batch_cull --set=gate_prepass=1 > gpurun_out/r04p_cull_pp.log 2>&1; grep -v amdgpu gpurun_out/r04p_cull_pp.log | head -12
timeout -k 10 200 python profiles/fusion_probe.py 3840x2160 8 --opt=batch_cull > gpurun_out/r04p_cull_shard.log 2>&1; grep -v amdgpu gpurun_out/r04p_cull_shard.log | head -4
for sz in 960x540 64x36; do timeout -k 10 200 python profiles/fusion_probe.py $sz 1 --auto --opt=batch_cull > gpurun_out/r04p_cull_$sz.log 2>&1; grep -v amdgpu gpurun_out/r04p_cull_$sz.log | head -4; done
