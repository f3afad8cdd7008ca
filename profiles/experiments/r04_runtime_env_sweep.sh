for e in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=1" "HSA_ENABLE_INTERRUPT=0"; do echo "== $e"; env $e timeout -k 10 120 python profiles/ab_probe.py --quick 2>&1 | grep "ms a frame"; done
echo "== default"; timeout -k 10 120 python profiles/ab_probe.py --quick 2>&1 | grep "ms a frame"
