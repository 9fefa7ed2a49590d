#!/usr/bin/env python3
"""Trim a rocprofv3 `--kernel-trace --stats` kernel_stats.csv to this repo's kernels.

usage: prof_summary.py <kernel_stats.csv> <out.csv>
torch's own kernels (input generation) are folded into one "other" line so the summary stays
readable; names are shortened to the kernel identifier."""
import csv
import re
import sys

OURS = re.compile(r"\(anonymous namespace\)::((?:bgr2gray|min_eigen|corner_[a-z]+|gaussian7|keypoint_border|rbrief|"
                  r"kdtree_[a-z]+|match_[a-z0-9]+|ransac_[a-z]+|fast_[a-z]+|pyr_[a-z0-9]+|grid_[a-z_]+|orb_[a-z]+|harris|"
                  r"ic_angle|retain_best)(?:_v4|_stream|_tiered|_lds|_rotate|_mfma)?_kernel)[<(]")


def main(src, dst):
    rows, other_ns, other_calls, total = [], 0, 0, 0
    with open(src) as f:
        for r in csv.DictReader(f):
            ns = int(r["TotalDurationNs"])
            total += ns
            m = OURS.search(r["Name"])
            if m:
                rows.append((m.group(1), int(r["Calls"]), ns, float(r["AverageNs"]), int(r["MinNs"]), int(r["MaxNs"])))
            else:
                other_ns += ns
                other_calls += int(r["Calls"])
    ours = sum(r[2] for r in rows)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct_of_vslam_kernels"])
        for r in sorted(rows, key=lambda r: -r[2]):
            w.writerow(list(r) + [f"{100.0 * r[2] / ours:.2f}"])
        w.writerow(["(torch/runtime kernels: synthetic input generation, copies)", other_calls, other_ns, "", "", "", ""])
    print(f"{dst}: {len(rows)} vslam kernels, {ours / 1e6:.2f} ms of {total / 1e6:.2f} ms traced")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
