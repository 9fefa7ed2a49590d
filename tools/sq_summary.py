#!/usr/bin/env python3
"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass
(SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
 + GRBM_GUI_ACTIVE, which lives in a counter block of its own).

usage: sq_summary.py <counter_collection.csv> <out.csv>
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md): the *_cycles columns are x 4.
GRBM_GUI_ACTIVE is summed over the 8 XCDs: a launch's duration in shader cycles is GRBM_GUI_ACTIVE / 8, whatever the clock was.

occupancy_waves_per_simd  = sum of the waves' lifetimes / (1024 SIMDs x the launch's cycles): how many waves a SIMD holds on
                            average while the kernel runs (north_star's "wave occupancy"; 8 is the hardware's limit).
valu_busy_frac            = share of the chip's vector-pipe time the kernel's vector instructions fill, CALIBRATED: the
                            kernel's SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE divided by the same ratio of pmc_calib_valu_kernel
                            (bench.py --pmc-calibrate), a kernel that keeps every vector pipe busy for its whole duration by
                            construction.  No clock and no cycles-per-instruction constant enters.  Without the calibration
                            kernel in the pass the column holds the uncalibrated estimate (4 cycles per counted quad-cycle,
                            1024 SIMDs) and `calibrated` says 0.
valu_busy_cycles_per_wave = 4 x SQ_ACTIVE_INST_VALU / waves, as before (bench.py combines it with ITS launch time)."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vslam_amd.profnames import kernel_id  # noqa: E402

SIMDS = 1024


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
    cal = agg.get("pmc_calib_valu_kernel")
    full = None   # SQ_ACTIVE_INST_VALU per GRBM_GUI_ACTIVE of a kernel whose vector pipes never idle
    if cal and cal["GRBM_GUI_ACTIVE"] > 0 and cal["SQ_ACTIVE_INST_VALU"] > 0:
        full = cal["SQ_ACTIVE_INST_VALU"] / cal["GRBM_GUI_ACTIVE"]
    rows = []
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
        wc, waves = v["SQ_WAVE_CYCLES"], v["SQ_WAVES"]
        if not wc or not waves:
            continue
        gui = v["GRBM_GUI_ACTIVE"]
        cycles = gui / 8.0 if gui > 0 else 0.0                      # all launches of the kernel, shader cycles
        occ = 4.0 * wc / (SIMDS * cycles) if cycles else ""
        if gui > 0 and full:
            busy, calibrated = v["SQ_ACTIVE_INST_VALU"] / gui / full, 1
        elif cycles:
            busy, calibrated = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (SIMDS * cycles), 0
        else:
            busy, calibrated = "", 0
        rows.append([k, int(waves / launches[k]), round(v["SQ_INSTS_VALU"] / waves), round(4 * wc / waves),
                     round(4 * v["SQ_ACTIVE_INST_VALU"] / waves),
                     round(100 * v["SQ_WAIT_ANY"] / wc, 1), round(100 * v["SQ_WAIT_INST_ANY"] / wc, 1),
                     round(100 * v["SQ_ACTIVE_INST_ANY"] / wc, 1), round(100 * v["SQ_ACTIVE_INST_VALU"] / wc, 1),
                     round(occ, 2) if occ != "" else "", round(busy, 3) if busy != "" else "", calibrated,
                     round(cycles / launches[k]) if cycles else ""])
    with open(dst, "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "waves_per_launch", "valu_insts_per_wave", "wave_cycles_per_wave_x4", "valu_busy_cycles_per_wave",
                    "wait_any_pct", "wait_inst_any_pct", "active_inst_any_pct", "active_inst_valu_pct",
                    "occupancy_waves_per_simd", "valu_busy_frac", "calibrated", "cycles_per_launch"])
        w.writerows(rows)
    print(open(dst).read())
    if full:
        import json
        with open(os.path.splitext(dst)[0] + "_calibration.json", "w") as o:
            json.dump({"kernel": "pmc_calib_valu_kernel: 8 waves per SIMD, independent v_fma_f32 only",
                       "sq_active_inst_valu_per_grbm_gui_active_at_full_pipes": full,
                       "nominal_ratio_4_cycles_per_quad_cycle_1024_simds_8_xcds": SIMDS / (8 * 4.0),
                       "valu_insts_per_wave": cal["SQ_INSTS_VALU"] / cal["SQ_WAVES"],
                       "cycles_per_valu_instruction_per_simd": cal["GRBM_GUI_ACTIVE"] / 8 * SIMDS / cal["SQ_INSTS_VALU"],
                       "what": "a SIMD issues one plain (non-packed) wave64 vector instruction about every 3.9 shader cycles even with "
                               "eight waves to choose from: the ceiling a kernel's vector-instruction issue rate is held against"}, o, indent=1)
        print(f"calibration: pmc_calib_valu_kernel has SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE = {full:.4f} "
              f"(= {full * 8 * 4 / SIMDS:.3f} of the nominal '4 cycles per quad-cycle on 1024 SIMDs'); "
              f"its {cal['SQ_INSTS_VALU'] / cal['SQ_WAVES']:.0f} vector instructions per wave took "
              f"{cal['GRBM_GUI_ACTIVE'] / 8 * SIMDS / (cal['SQ_INSTS_VALU']):.3f} cycles each per SIMD")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
