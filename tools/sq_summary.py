#!/usr/bin/env python3
"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass
(SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY).

usage: sq_summary.py <counter_collection.csv> <out.csv>"""
import collections
import csv
import re
import sys

NAME = re.compile(r"((?:bgr2gray|min_eigen|corner_[a-z]+|gaussian7|keypoint_border|rbrief|kdtree_[a-z]+|match_[a-z0-9]+|"
                  r"ransac_[a-z]+)(?:_v4|_stream|_tiered|_lds|_rotate|_mfma)?_kernel)")


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for r in csv.DictReader(open(src)):
        m = NAME.search(r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            launches[m.group(1)] += 1
    rows = []
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
        wc, waves = v["SQ_WAVE_CYCLES"], v["SQ_WAVES"]
        if not wc or not waves:
            continue
        rows.append([k, int(waves / launches[k]), round(v["SQ_INSTS_VALU"] / waves), round(4 * wc / waves),
                     round(100 * v["SQ_WAIT_ANY"] / wc, 1), round(100 * v["SQ_WAIT_INST_ANY"] / wc, 1),
                     round(100 * v["SQ_ACTIVE_INST_ANY"] / wc, 1), round(100 * v["SQ_ACTIVE_INST_VALU"] / wc, 1)])
    with open(dst, "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "waves_per_launch", "valu_insts_per_wave", "wave_cycles_per_wave_x4", "wait_any_pct",
                    "wait_inst_any_pct", "active_inst_any_pct", "active_inst_valu_pct"])
        w.writerows(rows)
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
