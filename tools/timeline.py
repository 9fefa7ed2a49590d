#!/usr/bin/env python3
"""Print the kernels of the last complete step of a rocprofv3 --kernel-trace CSV as a timeline (us from the step's
first kernel): queue, start, end, duration.  Usage: tools/timeline.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    if m and "at::" not in r["Kernel_Name"] and "pmc_calib" not in r["Kernel_Name"]:
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], m.group(1)))
ks.sort()
# a step starts at its first image kernel (cvtColor, or the detector that contains it) after a closing RANSAC kernel
FIRST = ("bgr2gray_kernel", "min_eigen_tiered_kernel", "min_eigen_kernel")
starts = [i for i, k in enumerate(ks) if k[3] in FIRST and i > 0 and ks[i - 1][3].startswith(("ransac_finish", "ransac_select", "ransac_mt"))]
s, e = starts[-2], starts[-1]
t0 = ks[s][0]
for st, en, q, name in ks[s:e + 1]:
    print("%-28s q%s  %8.1f -> %8.1f  (%7.1f)" % (name, q, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3))
