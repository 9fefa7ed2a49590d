#!/usr/bin/env python3
"""Trim a rocprofv3 `--kernel-trace --stats` kernel_stats.csv to this repo's kernels.

usage: prof_summary.py <kernel_stats.csv> <out.csv>
Every kernel of the library is listed (vslam_amd/profnames.py recognises them by their anonymous top-level namespace and the
`_kernel` suffix, not by a list of names); torch's own kernels (input generation) are folded into one "other" line."""
import csv
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vslam_amd.profnames import kernel_id  # noqa: E402


def main(src, dst):
    rows, other_ns, other_calls, total = {}, 0, 0, 0
    with open(src) as f:
        for r in csv.DictReader(f):
            ns = int(r["TotalDurationNs"])
            total += ns
            k = kernel_id(r["Name"])
            if k:   # template instances of one kernel are one row
                c, t, lo, hi = rows.get(k, (0, 0, 1 << 62, 0))
                rows[k] = (c + int(r["Calls"]), t + ns, min(lo, int(r["MinNs"])), max(hi, int(r["MaxNs"])))
            else:
                other_ns += ns
                other_calls += int(r["Calls"])
    ours = sum(v[1] for v in rows.values())
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct_of_vslam_kernels"])
        for k, (c, t, lo, hi) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, t, f"{t / c:.1f}", lo, hi, f"{100.0 * t / ours:.2f}"])
        w.writerow(["(torch/runtime kernels: synthetic input generation, copies)", other_calls, other_ns, "", "", "", ""])
    print(f"{dst}: {len(rows)} vslam kernels, {ours / 1e6:.2f} ms of {total / 1e6:.2f} ms traced")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
