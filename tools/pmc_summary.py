#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter unit KiB).

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv>
Applies the calibration MI355X_MICROARCH.md prescribes: the factor that makes the known-size
pmc_calib_copy{4,16}_kernel (1 GiB read, 1 GiB written) come out right (the same for 4 and 16 bytes per lane:
FETCH_SIZE x 2, WRITE_SIZE x 1).  Every kernel of the library is listed (vslam_amd/profnames.py)."""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vslam_amd.profnames import kernel_id  # noqa: E402


def load(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = kernel_id(r["Kernel_Name"])
            if not k:
                continue
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
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
            c = cal["pmc_calib_copy16_kernel" if "copy16" in k else "pmc_calib_copy4_kernel"]
            w.writerow([k, n[k], f"{fe[k]:.1f}", f"{wr.get(k, 0):.1f}", f"{c[0]:.3f}", f"{c[1]:.3f}",
                        f"{fe[k] * 1024 * c[0] / 1e6:.2f}", f"{wr.get(k, 0) * 1024 * c[1] / 1e6:.2f}"])
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
