#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter unit KiB).

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv>
Applies the calibration MI355X_MICROARCH.md prescribes: the factor that makes the known-size
pmc_calib_copy{4,16}_kernel (1 GiB read, 1 GiB written) come out right is applied per access width
(4-byte-per-lane kernels use the copy4 factor, 16-byte ones the copy16 factor)."""
import csv
import re
import sys
from collections import defaultdict

NAME = re.compile(r"((?:bgr2gray|min_eigen|corner_[a-z]+|gaussian7|keypoint_border|rbrief|kdtree_[a-z]+|match_[a-z0-9]+|"
                  r"ransac_[a-z]+|pmc_calib_copy\d+)(?:_v4|_stream|_tiered|_lds|_rotate|_mfma)?_kernel)")
WIDE = {}   # 16 B/lane side: write
# (calibration shows the same factors for 4 B and 16 B per lane: FETCH_SIZE x2, WRITE_SIZE x1)


def load(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            m = NAME.search(r["Kernel_Name"])
            if not m:
                continue
            tot[m.group(1)] += float(r["Counter_Value"])
            cnt[m.group(1)] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main(fetch_csv, write_csv, out):
    fe, n = load(fetch_csv)
    wr, _ = load(write_csv)
    gib = float(1 << 30)
    cal = {}
    for k in ("pmc_calib_copy4_kernel", "pmc_calib_copy16_kernel"):
        cal[k] = (gib / (fe[k] * 1024), gib / (wr[k] * 1024))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch",
                    "read_factor", "write_factor", "hbm_read_MB_per_launch", "hbm_write_MB_per_launch"])
        for k in sorted(fe, key=lambda k: -(fe[k] + wr.get(k, 0))):
            rf = cal["pmc_calib_copy16_kernel"][0] if WIDE.get(k) == "r" or "copy16" in k else cal["pmc_calib_copy4_kernel"][0]
            wf = cal["pmc_calib_copy16_kernel"][1] if WIDE.get(k) == "w" or "copy16" in k else cal["pmc_calib_copy4_kernel"][1]
            w.writerow([k, n[k], f"{fe[k]:.1f}", f"{wr.get(k, 0):.1f}", f"{rf:.3f}", f"{wf:.3f}",
                        f"{fe[k] * 1024 * rf / 1e6:.2f}", f"{wr.get(k, 0) * 1024 * wf / 1e6:.2f}"])
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
