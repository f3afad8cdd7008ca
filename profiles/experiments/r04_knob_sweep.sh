for ts in 0 8 16 32 64; do echo "== NDT_TRACE_TAIL_SOLO=$ts"; NDT_TRACE_TAIL_SOLO=$ts timeout -k 10 120 python profiles/ab_probe.py --quick 2>&1 | grep "ms a frame"; done
for sm in 1024 4096 16384; do echo "== NDT_TRACE_SMALL=$sm"; NDT_TRACE_SMALL=$sm timeout -k 10 120 python profiles/ab_probe.py --quick 2>&1 | grep "ms a frame"; done
