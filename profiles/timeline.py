#!/usr/bin/env python3
"""Print the kernel timeline of the last complete frame of a rocprofv3 --kernel-trace csv.
usage: timeline.py <kernel_trace.csv>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r['Kernel_Name'].split('(')[0][-34:], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
ks.sort(key=lambda x: x[1])
idx = [i for i, k in enumerate(ks) if 'k_frame_init' in k[0]]
i0, i1 = idx[-2], idx[-1]
t0 = ks[i0][1]
prev_end = t0
for k in ks[i0:i1]:
    print("%-36s start %8.1f us  dur %7.1f us  gap %6.1f us" % (k[0], (k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3, (k[1] - prev_end) / 1e3))
    prev_end = k[2]
print("frame span %.1f us" % ((ks[i1][1] - t0) / 1e3))
