#!/usr/bin/env python3
"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass
(SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY).

usage: sq_summary.py <counter_collection.csv> <out.csv>
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md): the *_cycles columns are x 4.
valu_busy_cycles_per_wave = the cycles a wave's vector instructions occupy its SIMD's vector pipe; summed over a launch's
waves and divided by (1024 SIMDs x launch duration x 2.4 GHz) it is the fraction of the chip's vector-pipe time the kernel
fills -- rocprof's VALUBusy, which bench.py reports beside the issue rate.  Every kernel of the library is listed."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vslam_amd.profnames import kernel_id  # noqa: E402


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for r in csv.DictReader(open(src)):
        k = kernel_id(r["Kernel_Name"])
        if not k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            launches[k] += 1
    rows = []
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
        wc, waves = v["SQ_WAVE_CYCLES"], v["SQ_WAVES"]
        if not wc or not waves:
            continue
        rows.append([k, int(waves / launches[k]), round(v["SQ_INSTS_VALU"] / waves), round(4 * wc / waves),
                     round(4 * v["SQ_ACTIVE_INST_VALU"] / waves),
                     round(100 * v["SQ_WAIT_ANY"] / wc, 1), round(100 * v["SQ_WAIT_INST_ANY"] / wc, 1),
                     round(100 * v["SQ_ACTIVE_INST_ANY"] / wc, 1), round(100 * v["SQ_ACTIVE_INST_VALU"] / wc, 1)])
    with open(dst, "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "waves_per_launch", "valu_insts_per_wave", "wave_cycles_per_wave_x4", "valu_busy_cycles_per_wave",
                    "wait_any_pct", "wait_inst_any_pct", "active_inst_any_pct", "active_inst_valu_pct"])
        w.writerows(rows)
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
