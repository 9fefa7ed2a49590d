#!/usr/bin/env python3
"""Throughput bench for the front-end hot path (BASELINE.json metric).

A step = one pass of extract(2 frames per pair) + match + RANSAC over one batch of synthetic
frame pairs already resident in HBM.  N = 1 runs BASELINE.json configs[2] (1280x720, 2000
keypoints, 4096 hypotheses, batch 256); with N > 1 every rank runs the same per-GPU batch on its
own shard (weak scaling, configs[3]) and the per-pair result records are gathered with the
library's own collective (vslam_gather_records: RCCL all-gather on the batch's stream).
Steps are handed round-robin to the contexts of a vslam_pipeline (--in-flight, default 4): each
batch in flight has its own frames, outputs and communicator; the timed region ends when every
batch is complete.  The data is SURVEY 8(d)'s regime (--data hard) since round 5.

Rank 0 prints ONE JSON line.  `roofline` is for the kernel with the largest share of the step,
timed with HIP events on the stream the kernels run on; `cpu_baseline` is the oracle (a CPU port
of the reference path, single thread) timed on a bounded sample of the same workload.

`python bench.py --gpus N` with N > 1 and no launcher in the environment starts the N ranks itself:
the parent (which never imports torch or touches a GPU) runs `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a CHILD process, relays rank 0's line and exits with the child's
code.  Under a launcher (RANK / WORLD_SIZE set, as the driver does for N > 1) it is a rank.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, keypoints, hypotheses, pairs per GPU)
    "C2": (640, 480, 1000, 1024, 64),
    "C3": (1280, 720, 2000, 4096, 256),
    "C5": (1920, 1080, 4000, 8192, 512),
    # the same step with the reference's OTHER extractor, extract_features(Frame&, nrows, ncols) (src/Frame.cpp:16-51: grid
    # ORB/FAST, the one north_star names; its only call is commented out at src/vslam.cpp:63): 4 x 4 cells, every keypoint it
    # finds (about 6800 per frame; "keypoints" = the slots per frame), then match + RANSAC as the live path.  Not the headline.
    "C3g": (1280, 720, 8192, 4096, 32),
}
SEED_INDEX = {"C2": 0, "C3": 1, "C5": 2, "C3g": 3}   # seeds of the BASELINE configs stay what they were before C3g existed
GRID = (4, 4)


def POSE_K(w, h):
    """The reference's camera matrix for a w x h frame (src/vslam.cpp:32)."""
    import numpy as np
    return np.array([[525.0, 0, w // 2], [0, 525.0, h // 2], [0, 0, 1]], np.float32)


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(kernel, w, h, K, H, M, gray_fused=False):
    """Compulsory HBM bytes PER FRAME PAIR for each kernel (DESIGN.md 'Roofline accounting'):
    every input read once, every output written once, temporaries not counted.
    gray_fused: cvtColor runs inside the detector's first kernel (no bgr2gray launch): that kernel then reads the
    3-byte image and writes the gray one."""
    px = w * h
    table = {
        "bgr2gray_kernel": 2 * (3 * px + px),
        # gray in (fused: BGR in, gray out); out: candidate keys, about 10 per kept keypoint, 8 B each
        # (the response image itself is not written on this path)
        "min_eigen_kernel": 2 * ((4 * px if gray_fused else px) + 80 * K),
        "corner_exact_kernel": 2 * (80 * K + (K * 8 // 5) * (25 + 8)),   # the list in; 5x5 gray windows of the ~1.6 K evaluated pixels in, their keys out
        "corner_select_kernel": 2 * (K * 8),
        "gaussian7_kernel": 2 * (px + px),
        "keypoint_border_kernel": 2 * (K * 8 * 2),
        "rbrief_kernel": 2 * (px + K * 8 + K * 32),   # the blurred image in (every 128 x 128 tile holds keypoints), keypoints in, descriptors out
        "kdtree_build_kernel": 2 * (K * 8 + K * 4),
        "match_knn2_kernel": 2 * K * 32 + K * 4,
        "match_compact_kernel": K * 4 + K * 8,
        "ransac_mt_kernel": 4 + H * 32,                # seed in, raw generator outputs out
        "ransac_sets_kernel": 2 * H * 32,              # raw outputs in, sets out
        "ransac_solve_kernel": H * 32 + M * 24 + H * 36,
        "ransac_score_kernel": H * 36 + M * 24 + H * 8,
        "ransac_count_kernel": H * 36 + M * 16 + H * 4 + H * 16,
        "ransac_rank_kernel": 8 * 36 + M * 24 + M * 16,
        "ransac_screen_kernel": H * 36 + 128 * 16 + H * 4,
        "ransac_cand_kernel": H * 4 + 8 * (36 + M * 16),
        "ransac_ties_kernel": H * 4 + H * 4,
        "ransac_tiesum_kernel": M * 24 + 36 + 4,          # per tied hypothesis; at least one per pair
        "ransac_select_kernel": H * 8 + M * 24 + M + M * 8 + 36,
        "ransac_finish_kernel": (H * 4 + H * 4) + (M * 24 + 36 + 4) + (H * 8 + M * 24 + M + M * 8 + 36),   # ties + one tie sum + select
    }
    return table.get(kernel, 0)


# Arithmetic ceilings for the kernels that are VALU-bound by construction: match and RANSAC.
#  * flops: SURVEY.md 8(d)'s algorithmic operation count per unit of work against the FP32 vector peak
#    (MI355X_MICROARCH.md: 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 32 lanes x 2 x 2.4 GHz);
#  * VALU issue: vector instructions the kernel actually issued (SQ_INSTS_VALU, rocprofv3 PMC pass of this
#    command, committed under profiles/ -- an instruction count per wave is a property of the code and the
#    workload, not of the run) x 2 cycles per wave instruction (a 64-lane wave on a SIMD-32) against
#    SIMDs x clock.  Both fractions are <= 1 by construction.
FP32_PEAK_TFLOPS = 157.3                       # FMA counted as 2: 256 CUs x 4 SIMDs x 32 lanes x 2 x 2.4 GHz
INT32_PEAK_TOPS = FP32_PEAK_TFLOPS / 2.0       # one 32-bit integer op per lane and clock
INT8_MFMA_PEAK_TOPS = 5000.0                   # dense int8 MFMA (MI355X_MICROARCH.md: about 2x the 2.5 PFLOP/s bf16 rate)
FP4_MFMA_PEAK_TOPS = 10000.0                   # dense FP4 / FP6 MFMA (MI355X_MICROARCH.md: about 10 PF; v_mfma_scale_f32_32x32x64_f8f6f4)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0          # G wave-instructions / s at the nominal 2 cycles per wave64 instruction (spec peak)


def valu_issue_ceiling():
    """The MEASURED issue ceiling (round 6): cycles a SIMD needs per plain vector instruction with eight waves of independent
    v_fma_f32 to choose from (tools/sq_summary.py, pmc_calib_valu_kernel in the committed SQ pass) -> G wave-instructions/s at
    2.4 GHz.  About 3.9 cycles, i.e. half the nominal rate: the spec's FP32 figure needs the packed forms."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_sq_counters_calibration.json")
    try:
        with open(path) as fh:
            c = json.load(fh)
        cyc = float(c["cycles_per_valu_instruction_per_simd"])
        return {"cycles_per_valu_instruction_per_simd": cyc, "G_wave_instructions_per_s_at_2.4GHz": SIMDS * CLOCK_GHZ / cyc,
                "source": os.path.relpath(path, ROOT)}
    except (OSError, ValueError, KeyError):
        return None
PROFILE_TAG = "r06"
SQ_PROFILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_sq_counters.csv")
PMC_PROFILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_hbm_traffic.csv")
STAMP_FILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_source_stamp.txt")
SOLVE_WORK = os.path.join(ROOT, "profiles", PROFILE_TAG + "_solve_work.json")


def solve_flops():
    """Floating-point operations of one compute_fundamental (src/RansacFilter.cpp:69-103) on the bench's data, COUNTED:
    tools/solve_flops.py has the oracle count Jacobi visits and rotations of both SVDs over 24 576 hypotheses and applies
    the per-visit / per-rotation operation counts written out there (profiles/r05_solve_work.json = round 4's count: the
    arithmetic has not changed; SURVEY 8(d) guessed 3000)."""
    try:
        with open(SOLVE_WORK) as fh:
            return float(json.load(fh)["flop_per_hypothesis"])
    except (OSError, KeyError, ValueError):
        return 14850.0


ALG_OPS = {
    # kernel: (what one unit is, algorithmic ops per unit, kind, peak in Tops/s)
    # ransac_count_kernel has no entry: it decides most (hypothesis, match) pairs without evaluating them (bail-out),
    # so SURVEY's H x M x 40 flop is not work it performs and a flop fraction of it would mean nothing
    "ransac_score_kernel": ("(hypothesis, match) residual evaluations", 40.0, "flop", FP32_PEAK_TFLOPS),
    "ransac_solve_kernel": ("hypotheses (A, two Jacobi SVDs, F = U diag Vt: counted per visit and rotation, tools/solve_flops.py -> "
                            "profiles/" + PROFILE_TAG + "_solve_work.json; about half of them f64)", solve_flops(), "flop", FP32_PEAK_TFLOPS),
    "min_eigen_kernel": ("pixels (stencil work, about 60 int/flop per pixel: SURVEY.md 8d)", 60.0, "flop", FP32_PEAK_TFLOPS),
    # the matcher forms each 256-bit Hamming distance as an FP4 (+-1) dot product on the matrix cores: 256 multiply-adds
    # (a stream of nothing but these multiplies issues at 7.25 Pop/s on this chip: tools/mfma_fp4_probe.hip)
    "match_knn2_kernel": ("(query, train) descriptor pairs", 512.0, "FP4 op (256 multiply-adds on MFMA)", FP4_MFMA_PEAK_TOPS),
}


def kernel_sources():
    """{timing slot: source file} for every __global__ kernel under vslam_amd/csrc."""
    import re
    from vslam_amd.profnames import slot_of
    out = {}
    d = os.path.join(ROOT, "vslam_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith(".hip"):
            with open(os.path.join(d, name)) as fh:
                for k in re.findall(r"__global__[^;{]*?\b([a-z][a-z0-9_]*_kernel)\s*\(", fh.read(), flags=re.S):
                    out[slot_of(k)] = name
    return out


def source_stamp():
    """One hash per kernel source (+ one over the shared headers): instruction counts per wave and HBM traffic per launch are
    properties of a BUILD, so a kernel's committed counter rows (tools/prof_all.sh writes these stamps beside them) describe
    the running library exactly when the stamp of the file that defines the kernel, and of the headers, still agree."""
    import hashlib
    d = os.path.join(ROOT, "vslam_amd", "csrc")
    out, hdr = {}, hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if not name.endswith((".h", ".hip")):
            continue
        with open(os.path.join(d, name), "rb") as fh:
            data = fh.read()
        if name.endswith(".h"):
            hdr.update(name.encode() + b"\0" + data)
        elif name.endswith(".hip"):
            out[name] = hashlib.sha256(data).hexdigest()[:16]
    out["headers"] = hdr.hexdigest()[:16]
    return out


_STAMP_CACHE = {}


def counters_current(kernel=None):
    """Do the committed counter rows of `kernel` (a timing slot; None: of every kernel) come from this build's sources?"""
    if "now" not in _STAMP_CACHE:
        try:
            with open(STAMP_FILE) as fh:
                _STAMP_CACHE["then"] = json.load(fh)
        except (OSError, ValueError):
            _STAMP_CACHE["then"] = {}
        _STAMP_CACHE["now"] = source_stamp()
        _STAMP_CACHE["src"] = kernel_sources()
    then, now = _STAMP_CACHE["then"], _STAMP_CACHE["now"]
    if not then or then.get("headers") != now["headers"]:
        return False
    if kernel is None:
        return then == now
    f = _STAMP_CACHE["src"].get(kernel)
    return f is not None and then.get(f) == now.get(f)


def _profile_row(path, kernel, weight):
    """The committed counter row of the kernel that runs under timing slot `kernel`; where several variants share a slot
    (the pool's idle min_eigen_v4 launches beside min_eigen_tiered) the one with the largest `weight` column."""
    import csv
    from vslam_amd.profnames import slot_of
    if not os.path.exists(path) or not counters_current(kernel):
        return None
    best = None
    with open(path) as f:
        for r in csv.DictReader(f):
            if slot_of(r["kernel"]) == kernel and (best is None or float(r[weight]) > float(best[weight])):
                best = r
    return best


def sq_counters(kernel):
    """(waves per launch, VALU instructions per wave, VALU-busy cycles per wave) from the committed rocprofv3 SQ summary."""
    r = _profile_row(SQ_PROFILE, kernel, "valu_insts_per_wave")
    if r is None:
        return None
    return float(r["waves_per_launch"]), float(r["valu_insts_per_wave"]), float(r.get("valu_busy_cycles_per_wave") or 0.0)


def sq_profiled(kernel):
    """What the committed SQ pass says about the kernel AS PROFILED (numerator and denominator from the same launches, nothing
    of this run's timing in it): wave occupancy (mean resident waves per SIMD over the launch: sum of the waves' lifetimes /
    (1024 SIMDs x GRBM_GUI_ACTIVE / 8)) and the vector pipes' busy share, calibrated against pmc_calib_valu_kernel -- a kernel
    that keeps every vector pipe busy by construction -- instead of a cycles-per-instruction constant (tools/sq_summary.py)."""
    r = _profile_row(SQ_PROFILE, kernel, "valu_insts_per_wave")
    if r is None or not r.get("occupancy_waves_per_simd"):
        return None
    out = {"occupancy_waves_per_simd": float(r["occupancy_waves_per_simd"])}
    if r.get("valu_busy_frac"):
        out["valu_busy_frac"] = float(r["valu_busy_frac"])
        out["valu_busy_calibrated"] = r.get("calibrated") == "1"
    return out


SIMDS, CLOCK_GHZ = 256 * 4, 2.4


def arithmetic_view(kernel, units, ms_per_launch, full_batch):
    """flops / issue fractions of one launch of `kernel` that processed `units` units of work."""
    if ms_per_launch <= 0:
        return None
    t = ms_per_launch * 1e-3
    view = {}
    if kernel in ALG_OPS:
        what, ops, kind, peak = ALG_OPS[kernel]
        tops = units * ops / t / 1e12
        view = {"unit_of_work": what, "units_per_launch": units, "ops_per_unit": ops, "op_kind": kind,
                "achieved": tops, "peak": peak, "unit": "Tops/s (vector ALU peak for this kind of op)",
                "frac": tops / peak}
    sq = sq_counters(kernel) if full_batch else None
    if sq:
        waves, insts, busy = sq
        ginst = waves * insts / t / 1e9
        view["valu_issue"] = {"achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G VALU wave-instructions/s",
                              "frac": ginst / VALU_PEAK_GINST, "waves_per_launch": waves, "valu_insts_per_wave": insts,
                              "source": os.path.relpath(SQ_PROFILE, ROOT)}
        ceil = valu_issue_ceiling()
        if ceil:
            view["valu_issue"]["measured_ceiling"] = ceil
            view["valu_issue"]["frac_of_measured_ceiling"] = ginst / ceil["G_wave_instructions_per_s_at_2.4GHz"]
        prof = sq_profiled(kernel)
        if prof and "valu_busy_frac" in prof:
            # the share of the chip's vector-pipe time the kernel's instructions fill, as profiled: SQ_ACTIVE_INST_VALU per
            # GRBM_GUI_ACTIVE relative to the same ratio of a kernel whose pipes never idle (no clock, no cycles-per-instruction
            # constant, nothing of this run's timing: cannot exceed 1 by construction, so nothing is clamped)
            view["valu_busy"] = {"frac": prof["valu_busy_frac"], "calibrated": prof["valu_busy_calibrated"],
                                 "what": "SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE of the kernel over the same ratio of pmc_calib_valu_kernel "
                                         "(eight waves per SIMD of independent v_fma_f32: vector pipes busy throughout), one rocprofv3 pass"}
        elif busy > 0:
            frac = waves * busy / (SIMDS * t * CLOCK_GHZ * 1e9)
            view["valu_busy"] = {"frac": frac, "calibrated": False, "valu_busy_cycles_per_wave": busy,
                                 "what": "UNCALIBRATED: SQ_ACTIVE_INST_VALU x 4 summed over the launch's waves / (1024 SIMDs x this run's "
                                         "launch time x 2.4 GHz)"}
        if prof:
            view["occupancy_waves_per_simd"] = prof["occupancy_waves_per_simd"]
    return view or None


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/, C3 batch of 256 pairs);
    bench.py cannot collect PMC counters on itself."""
    import csv
    from vslam_amd.profnames import slot_of
    if _profile_row(PMC_PROFILE, kernel, "FETCH_SIZE_KiB_per_launch") is None:
        return None
    total = 0.0   # every kernel accounted under the slot (the matcher's spreading pre-pass, the pool's idle launches)
    with open(PMC_PROFILE) as f:
        for r in csv.DictReader(f):
            if slot_of(r["kernel"]) == kernel:
                total += (float(r["hbm_read_MB_per_launch"]) + float(r["hbm_write_MB_per_launch"])) * 1e6
    return total


def _cpu_worker(args):
    wl, n_pairs, seed = args
    return cpu_baseline(wl, n_pairs, seed)["value"]


def cpu_baseline_all_cores(wl, pairs_per_proc, seed):
    """The same oracle as independent pairs on every host core: one process per core."""
    import multiprocessing as mp
    procs = max(1, min(len(os.sched_getaffinity(0)), 16))   # a one-GPU box's CPU share is 16 cores
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        rates = pool.map(_cpu_worker, [(wl, pairs_per_proc, seed + 17 * i) for i in range(procs)])
    dt = time.perf_counter() - t0
    return {"value": float(sum(rates)), "unit": "frame-pairs/s", "cores": procs, "kind": "port",
            "sample": f"{procs} concurrent processes x {pairs_per_proc} pairs of {wl}, oracle; sum of per-process rates "
                      f"(input generation excluded), {dt:.1f} s wall"}


def cpu_baseline(wl, sample_pairs, seed):
    """The oracle (oracle/, a CPU port of the reference path; the reference itself cannot be built
    here) on `sample_pairs` pairs of the same workload, one thread.  (The all-cores figure's worker; the single-thread
    figure of the line comes from oracle_on_frames, on the timed batch's own bytes.)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    from vslam_amd import synth
    w, h, K, H, _ = WORKLOADS[wl]
    o = Oracle()
    bgr = synth.frames_numpy(seed, sample_pairs, w, h)
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    t0 = time.perf_counter()
    for p in range(sample_pairs):
        a = o.extract_features(bgr[p], K, ca, sa, pat)
        b = o.extract_features(bgr[sample_pairs + p], K, ca, sa, pat)
        o.match_features(a["xy"], a["desc"], b["xy"], b["desc"], seed ^ p, H, 10.0)
    dt = time.perf_counter() - t0
    return {"value": sample_pairs / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{sample_pairs} pairs of {wl} ({w}x{h}, {K} kp, {H} hyp), oracle single thread, {dt:.1f} s"}


def oracle_on_frames(frames_a, frames_b, seeds, K, H, thr, gpu_out, pair0, what):
    """Run the oracle on the very frames the GPU step processed (copied back from the device) and hold the step's
    output to it: keypoint counts of both frames, the number of inlier matches, the match list and F bit for bit.
    Returns (cpu_baseline dict, parity dict); the time is the oracle's own (copies excluded)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    from vslam_amd import synth
    o = Oracle()
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    n = frames_a.shape[0]
    P = gpu_out["n"].shape[0] // 2
    bad = []
    t0 = time.perf_counter()
    for p in range(n):
        a = o.extract_features(frames_a[p], K, ca, sa, pat)
        b = o.extract_features(frames_b[p], K, ca, sa, pat)
        ref = o.match_features(a["xy"], a["desc"], b["xy"], b["desc"], int(seeds[p]) & 0xFFFFFFFF, H, thr)
        g = pair0 + p
        k = len(ref["matches"])
        ok = (int(gpu_out["n"][g]) == a["n"] and int(gpu_out["n"][P + g]) == b["n"] and int(gpu_out["best"][g, 3]) == k
              and np.array_equal(gpu_out["matches"][g, :k], ref["matches"])
              and np.array_equal(gpu_out["F"][g].view(np.uint32), np.asarray(ref["F"], np.float32).reshape(-1).view(np.uint32)))
        if not ok:
            bad.append(g)
    dt = time.perf_counter() - t0
    h, w = frames_a.shape[1:3]
    base = {"value": n / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n} pairs of the timed batch itself ({what}: {w}x{h}, {K} kp, {H} hyp) copied back from the device, "
                      f"oracle single thread, {dt:.1f} s"}
    parity = {"pairs": n, "bit_exact": not bad, "checked": "keypoint counts, inlier-match lists and F (as uint32) of the timed step's output vs the oracle on the same bytes"}
    if bad:
        parity["mismatching_pairs"] = bad[:16]
    return base, parity


def grid_algorithmic_bytes(scope, w, h, kp):
    """Compulsory HBM bytes PER FRAME of the grid extractor's stages (4 x 4 cells, levels 0-5 of an 8-level 1.2 pyramid hold
    keypoints at 1280x720): inputs read once, outputs written once."""
    px = w * h
    lv = [1.0 / (1.2 ** (2 * l)) for l in range(6)]
    pyr = px * sum(lv[1:])                      # one pyramid's levels 1..5
    inner = sum(max(0.0, (w / 4 / 1.2 ** l - 62)) * max(0.0, (h / 4 / 1.2 ** l - 62)) for l in range(6)) * 16
    return {"grid_outline_gray_kernel": 3 * px + px,
            "grid_pyramid_kernels": 2 * (px * sum(lv[:5]) + pyr),          # frame + cell pyramids: level l - 1 in, level l out
            "fast_collect_kernel": inner * (1 + 0.06 * 4),                    # inner regions in, list entries out
            "orb_select_kernels": kp * 3 * (4 + 4 + 81),                      # entries + responses + 9x9 windows
            "grid_assemble_kernel": kp * (4 + 16),
            "orb_compute_kernels": 2 * (px + pyr) + kp * (961 + 512 + 16 + 48)}.get(scope, 0.0)


def grid_workload(args):
    """--workload C3g: the bench step with the grid ORB/FAST extractor in front (both frames of every pair), then the live
    path's match + RANSAC on its keypoints.  Same contract as the main line; one GPU only (a secondary workload)."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    from vslam_amd import capi, shard, synth
    from vslam_amd.capi import Pipeline
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w, h, K, H, P = WORKLOADS["C3g"]
    if args.pairs:
        P = args.pairs
    thr = 10.0
    seed = 0x5EED0000 + SEED_INDEX["C3g"]
    n_slots = max(1, min(16, args.in_flight))
    pipe = Pipeline(0, n_slots)
    ctx = pipe.contexts[0]
    make_frames = {"easy": synth.frames_torch, "hard": synth.frames_torch_hard, "photo": synth.frames_torch_photo}[args.data]
    pat = torch.from_numpy(synth.brief_pattern()).to(dev)

    class Slot:
        pass
    slots = []
    for s_ in range(n_slots):
        sl = Slot()
        sl.bgr = make_frames(seed + 7919 * s_, P, w, h, dev)
        sl.seeds_np = shard.pair_seeds(seed, s_ * P, (s_ + 1) * P)
        sl.seeds = torch.from_numpy(sl.seeds_np.view(np.int32)).to(dev)
        sl.g = dict(xy=torch.zeros((2 * P, K, 2), dtype=torch.float32, device=dev),
                    desc=torch.zeros((2 * P, K, 32), dtype=torch.uint8, device=dev),
                    angle_octave=torch.zeros((2 * P, K, 2), dtype=torch.float32, device=dev),
                    n=torch.zeros((2 * P,), dtype=torch.int32, device=dev))
        sl.m = dict(matches=torch.zeros((P, K, 2), dtype=torch.int32, device=dev), best=torch.zeros((P, 4), dtype=torch.int32, device=dev),
                    F=torch.zeros((P, 9), dtype=torch.float32, device=dev), prelim_m=torch.zeros((P,), dtype=torch.int32, device=dev))
        sl.used = 0
        slots.append(sl)
    torch.cuda.synchronize(dev)

    def run(c, sl):
        g = c.extract_features_grid(sl.bgr, GRID[0], GRID[1], pat, K, out=sl.g)   # draws the cell outlines into sl.bgr (:32), every step
        c.match_features(g["xy"][:P], g["desc"][:P], g["n"][:P], g["xy"][P:], g["desc"][P:], g["n"][P:], sl.seeds, H, thr, out=sl.m)

    def step(k):
        sl = slots[k % n_slots]
        t, c = pipe.acquire()
        run(c, sl)
        pipe.commit(t)
        sl.used += 1

    if args.pmc_calibrate:   # the known-size copies and the busy-pipe kernel the counter summaries calibrate against
        a_ = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
        b_ = torch.empty_like(a_)
        ctx.debug_stream_copy(a_, b_, 4)
        ctx.debug_stream_copy(a_, b_, 16)
        ctx.debug_valu_calib()
        ctx.synchronize()
        del a_, b_
    if args.alone:
        for c_ in pipe.contexts:
            c_.prof_enable(True)
    for k in range(n_slots):
        step(k)
    pipe.drain()
    for k in range(args.warmup):
        step(k)
    pipe.drain()
    for sl in slots:
        sl.used = 0
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    pipe.drain()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    ms_step = dt / args.steps * 1e3
    used = [sl for sl in slots if sl.used > 0]
    for sl in used:
        sl.host = {k: v.cpu().numpy() for k, v in list(sl.g.items()) + list(sl.m.items())}
        assert (sl.host["n"] > 1000).all() and (sl.host["best"][:, 0] >= 0).all() and (sl.host["best"][:, 3] >= 8).all(), "bench output degenerate"
    n_kp = np.concatenate([sl.host["n"] for sl in used])
    best = np.concatenate([sl.host["best"] for sl in used])
    data_label = {"easy": "easy data: translated texture + one moving block",
                  "hard": "SURVEY 8(d) data: rotation + parallax, sub-pixel resampling",
                  "photo": "photographic data: windows of four public-domain photographs under small camera motions"}[args.data]
    result = {
        "metric": "frame-pairs/sec (grid ORB/FAST extract+match+RANSAC) @1280x720, 4x4 cells, 4096 hyp; secondary workload, not BASELINE.json's headline",
        "value": P * args.steps / dt, "unit": "frame-pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64",
        "data": "synthetic (" + data_label + ")",
        "config": {"workload": f"C3g: {w}x{h}, grid ORB/FAST extractor (src/Frame.cpp:16-51) {GRID[0]}x{GRID[1]} cells on both frames of a pair, "
                               f"all its keypoints ({K} slots per frame), {H} hypotheses, batch {P} pairs per GPU, {data_label}",
                   "pairs_per_gpu": P, "batches_in_flight": n_slots, "parallelism": f"one GPU; {n_slots} batch(es) in flight (vslam_pipeline_*)"},
        "setup_steps": n_slots, "mean_keypoints": float(n_kp.mean()), "mean_inlier_matches": float(best[:, 3].mean()),
        "workspace_bytes": pipe.workspace_bytes(), "workspace_bytes_per_context": ctx.workspace_bytes(),
    }
    exit_code = 0
    if not args.no_profile_pass:   # per-stage times, HIP events on the stream, one context, one batch after the other
        s0 = slots[0]
        psteps = 3
        # the extractor alone (what tools/grid_bench.py times)
        ctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            ctx.extract_features_grid(s0.bgr, GRID[0], GRID[1], pat, K, out=s0.g)
        ctx.synchronize()
        ext_ms = (time.perf_counter() - t1) / 10 * 1e3
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(psteps):
            run(ctx, s0)
        rep = ctx.prof_report()
        ctx.prof_enable(False)
        mean_kp = float(n_kp.mean())
        ks = []
        for name, (ms, cnt) in rep.items():
            per = ms / max(cnt, 1)
            alg = grid_algorithmic_bytes(name, w, h, mean_kp) * 2 * P
            k = {"kernel": name, "ms_per_launch": per, "launches_per_step": cnt / psteps}
            if alg:
                k["alg_bytes_per_launch"] = alg
                k["alg_GBps"] = alg / (per * 1e-3) / 1e9 if per > 0 else 0.0
            ks.append(k)
        ks.sort(key=lambda k: -k["ms_per_launch"] * k["launches_per_step"])
        result["kernels"] = ks
        result["grid_extractor"] = {"ms_per_call": ext_ms, "frames": 2 * P, "us_per_frame": ext_ms / (2 * P) * 1e3,
                                    "what": "vslam_extract_features_grid alone on the batch's 2 x pairs frames, one context, 10 calls; "
                                            "the stages inside it are the grid_* / fast_* / orb_* rows of `kernels` (HIP events around each stage)"}
        gk = [k for k in ks if "alg_bytes_per_launch" in k]
        if gk:
            top = gk[0]
            result["roofline"] = {"kernel": top["kernel"], "bound": "hbm", "achieved": top["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": top["alg_GBps"] / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": top["ms_per_launch"],
                                  "note": "stage of the grid extractor with the largest share of the step; algorithmic bytes as "
                                          "grid_algorithmic_bytes() states them; the FAST stage is bound by its vector arithmetic, not by bandwidth"}
    # parity of what was timed + CPU baseline: the oracle on the very frames the device processed
    n_chk = min(P, args.cpu_pairs if args.cpu_pairs > 0 else 2, 8)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    o = Oracle()
    patn = synth.brief_pattern()
    sl = used[0]
    fa, fb = sl.bgr[:n_chk].cpu().numpy(), sl.bgr[P:P + n_chk].cpu().numpy()
    bad = []
    t1 = time.perf_counter()
    for p_ in range(n_chk):
        ra = o.extract_features_grid(fa[p_], GRID[0], GRID[1], patn)
        rb = o.extract_features_grid(fb[p_], GRID[0], GRID[1], patn)
        ref = o.match_features(ra[1], ra[2], rb[1], rb[2], int(sl.seeds_np[p_]) & 0xFFFFFFFF, H, thr)
        na, nb, k = len(ra[1]), len(rb[1]), len(ref["matches"])
        ho = sl.host
        ok = (int(ho["n"][p_]) == na and int(ho["n"][P + p_]) == nb and np.array_equal(ho["xy"][p_, :na].view(np.uint32), ra[1].view(np.uint32))
              and np.array_equal(ho["desc"][p_, :na], ra[2]) and np.array_equal(ho["desc"][P + p_, :nb], rb[2])
              and int(ho["best"][p_, 3]) == k and np.array_equal(ho["matches"][p_, :k], ref["matches"])
              and np.array_equal(ho["F"][p_].view(np.uint32), np.asarray(ref["F"], np.float32).reshape(-1).view(np.uint32)))
        if not ok:
            bad.append(p_)
    secs = time.perf_counter() - t1
    result["parity_in_bench"] = {"pairs": n_chk, "bit_exact": not bad,
                                 "checked": "keypoint counts, coordinates and descriptors of both frames (grid extractor), inlier-match lists and F (as "
                                            "uint32) of the timed steps' output vs the oracle on the same bytes"}
    if bad:
        result["parity_in_bench"]["mismatching_pairs"] = bad
        exit_code = 3
    if args.cpu_pairs > 0:
        result["cpu_baseline"] = {"value": n_chk / secs, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                  "sample": f"{n_chk} pairs of the timed batch copied back from the device, oracle single thread, {secs:.1f} s"}
    pipe.close()
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(result) + "\n").encode())
    os.close(real_stdout)
    return exit_code


def self_launch(n_gpus, argv, limit_s):
    """--gpus N > 1 without a launcher: start the ranks as a CHILD process tree (never exec: this parent has
    not imported torch or touched the GPU, and it stays alive to relay the result).  Rank 0's JSON line is the
    only thing the ranks write to stdout.  The tree gets a wall-clock limit: a rank stuck in a rendezvous or a
    collective would otherwise hang the caller; on expiry the whole process group is ended (TERM, then KILL) and
    the exit code is non-zero with one line of reason on stderr."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)   # its own process group
    try:
        out, _ = proc.communicate(timeout=limit_s)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 5)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        try:
            out, _ = proc.communicate(timeout=5)
        except Exception:
            out = ""
        print(f"bench.py: the {n_gpus} ranks did not finish within {limit_s:.0f} s (--launch-timeout): process group ended", file=sys.stderr)
        rc = 124
    for line in (out or "").splitlines():
        # the contract is ONE JSON line on stdout; anything else a library printed there (gloo's connection notes)
        # goes to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return rc


def dry_run(args, rank, world):
    """VSLAM_BENCH_DRY=1: the launcher, process group, record gather and max-over-ranks timing with made-up
    records and NO kernels (no GPU needed) -- what tests/test_bench_launch.py runs on a CPU-only box.  The line it
    prints carries no measurement (value null, metric says so)."""
    sys.stdout.flush()
    real_stdout = os.dup(1)   # one line on stdout: see main()
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    from vslam_amd import shard
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    if os.environ.get("VSLAM_BENCH_DRY_SLEEP"):   # a rank that hangs (tests/test_bench_launch.py: the launcher's wall-clock limit)
        time.sleep(float(os.environ["VSLAM_BENCH_DRY_SLEEP"]))
    _, _, K, _, P = WORKLOADS[args.workload]
    P = args.pairs or 4
    lo, hi = shard.shard_range(world * P, rank, world)
    g = torch.Generator().manual_seed(99)
    F_all = torch.randn((world * P, 9), generator=g)
    best_all = torch.randint(-1, K, (world * P, 4), generator=g, dtype=torch.int32)
    m_all = torch.randint(0, K, (world * P, K, 2), generator=g, dtype=torch.int32)
    seeds = shard.pair_seeds(0x5EED0000, lo, hi)
    assert seeds.shape[0] == P and int(seeds[0]) == (0x5EED0000 ^ lo)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = shard.pack_records(F_all[lo:hi], best_all[lo:hi], m_all[lo:hi])
        out = shard.gather_records(rec, world, n_items=world * P)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    all_t = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(all_t, t)
    rank_ms = [float(x.item()) / max(args.steps, 1) * 1e3 for x in all_t]
    F, best, m = shard.unpack_records(out, K)
    ok = torch.equal(F.view(torch.int32), F_all.view(torch.int32)) and torch.equal(best, best_all) and torch.equal(m, m_all)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    line = json.dumps({"metric": "DRY RUN: launcher + record gather only, no kernels, not a measurement", "value": None,
                       "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                       "ms_per_step": None, "gather_ok": bool(flag.item()), "data": "synthetic",
                       "per_rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "ranks": rank_ms,
                                                "note": "dry run: gather time only"},
                       "config": {"workload": "dry", "pairs_per_gpu": P, "parallelism": f"pairs sharded x{world}, gloo all_gather of result records"}})
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (line + "\n").encode())
    os.close(real_stdout)
    return 0 if flag.item() else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU and batch (default: the workload's batch)")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="batches in flight per GPU: contexts of the vslam_pipeline the steps are handed to round-robin "
                         "(1 = one batch after the other on one context, the arrangement of rounds 1-4)")
    ap.add_argument("--cpu-pairs", type=int, default=150,
                    help="pairs of the timed batches that the oracle recomputes on the host, spread over the contexts: the CPU baseline "
                         "and the in-bench parity check at once (0 = no CPU leg; the parity check then still covers 4 pairs)")
    ap.add_argument("--data", default="hard", choices=["easy", "hard", "photo"],
                    help="hard (default since round 5) = SURVEY 8(d)'s regime: rotation + parallax, sub-pixel resampling, 40-45 %% outlier "
                         "matches (synth.frames_torch_hard); easy = translated texture + one moving block (the headline's data of rounds 1-4); "
                         "photo = windows of the four public-domain photographs of tests/golden/real_v1.npz under small camera motions "
                         "(synth.frames_torch_photo): real image statistics, a secondary regime")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (other data regime, full-evaluation worst case, in-flight sweep, C2 / C5)")
    ap.add_argument("--cpu-all-cores-pairs", type=int, default=24,
                    help="pairs per process for the all-host-cores CPU figure (0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--width", type=int, default=0,
                    help="frame width instead of the workload's (e.g. 1278: a width that is no multiple of 4 runs on padded internal rows); "
                         "the default run reports one such shape under other_shapes")
    ap.add_argument("--pose", action="store_true",
                    help="every step also runs the pose stages behind the path (vslam_frontend_pairs_pose: extract_Rt, triangulate, "
                         "reprojection filter); the default run reports this arrangement under pose_chain.in_flight")
    ap.add_argument("--alone", action="store_true",
                    help="for rocprofv3 passes: every kernel of a step on ONE stream, nothing beside it (k-d build, blur and generator in "
                         "line), so that a traced kernel's duration is its own -- the arrangement of the per-kernel HIP-event pass; use with "
                         "--in-flight 1")
    ap.add_argument("--solver", default="exact", choices=["exact", "gram"],
                    help="gram = the opt-in MFMA / normal-matrix 8-point solver (VSLAM_OPT_RANSAC_SOLVER 1): NOT bit-exact, "
                         "never the headline number; the line is labelled")
    ap.add_argument("--comm", default="per-rank", choices=["per-rank", "per-context"],
                    help="N > 1: RCCL communicators for the record gather.  per-rank (default): ONE communicator per rank, the gathers of "
                         "the batches in flight issued on it in ticket order, each on its batch's stream; per-context: one communicator per "
                         "batch in flight (its collectives can overlap those of the other batches)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("VSLAM_BENCH_LAUNCH_TIMEOUT", "1500")),
                    help="wall-clock limit in seconds for the ranks this process starts itself (--gpus N > 1 without a launcher)")
    ap.add_argument("--pmc-calibrate", action="store_true",
                    help="also run two 1 GiB streaming copies (4 B and 16 B per lane) so FETCH_SIZE/WRITE_SIZE can be calibrated")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (child processes; see self_launch)
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))
    if os.environ.get("VSLAM_BENCH_DRY"):
        sys.exit(dry_run(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))))
    if args.workload == "C3g":
        if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            sys.exit("bench.py: C3g (grid extractor workload) is a one-GPU secondary measurement")
        sys.exit(grid_workload(args))

    # The contract is ONE line on stdout.  RCCL prints a version banner there when its communicator comes up and gloo its
    # connection notes, from C code: hand every such write to stderr by pointing fd 1 at it for the whole run, and write
    # the result line to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from vslam_amd import capi, shard, synth
    from vslam_amd.capi import Pipeline

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # VSLAM_BENCH_FORCE_DIST=1 takes the N > 1 code path (process group, communicators, record gather, max-over-ranks
    # timing) with a single rank, which is how that path is exercised with RCCL on a one-GPU box
    multi = world > 1 or bool(os.environ.get("VSLAM_BENCH_FORCE_DIST"))
    # The record gather -- the only exchange on the path -- goes through the PRODUCT's collective: vslam_comm_* /
    # vslam_gather_records (RCCL all-gather on the context's stream, include/vslam_amd.h).  torch.distributed is the control
    # plane only (the unique ids, the barriers around the timed region, the per-rank times) and runs over gloo, so the one
    # RCCL communicator per context in this process is the library's.  VSLAM_BENCH_GATHER=host rehearses N > 1 on a box with
    # fewer GPUs than ranks (RCCL refuses two ranks on one device): the records then travel through host memory over gloo.
    gather_mode = os.environ.get("VSLAM_BENCH_GATHER", "rccl") if multi else "none"
    if args.gpus > 1 or multi:
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("VSLAM_BENCH_BACKEND", "gloo"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    w, h, K, H, P = WORKLOADS[args.workload]
    if args.pairs:
        P = args.pairs
    if args.width:
        w = args.width
    thr = 10.0                                    # RansacFilter rf(8, 100, 10), src/vslam.cpp:19
    seed = 0x5EED0000 + SEED_INDEX[args.workload]
    n_slots = max(1, min(16, args.in_flight))
    pipe = Pipeline(local_rank, n_slots)
    if args.solver == "gram":
        pipe.set_option(capi.Context.OPT_RANSAC_SOLVER, 1)
    ctx = pipe.contexts[0]
    FRAME_MAKERS = {"easy": synth.frames_torch, "hard": synth.frames_torch_hard, "photo": synth.frames_torch_photo}
    make_frames = FRAME_MAKERS[args.data]
    pat = torch.from_numpy(synth.brief_pattern()).to(dev)
    ca, sa = synth.keypoint_rotation()
    lo, hi = shard.shard_range(world * P, rank, world)          # this rank's slice of a global batch
    words = shard.record_words(K)

    # One set per batch in flight: its own frames (different content per context and rank: nothing is shared between the
    # batches the timed loop hands out), seeds by GLOBAL pair index, outputs, record buffers, communicator.
    class Slot:
        pass
    slots = []
    n_comms = (1 if args.comm == "per-rank" else n_slots) if gather_mode == "rccl" else 0
    uids = [None] * n_comms
    comms = []
    if gather_mode == "rccl":
        if rank == 0:
            uids = [capi.comm_unique_id() for _ in range(n_comms)]
        dist.broadcast_object_list(uids, src=0)               # 128 bytes per communicator through the torch store
        # collective: same order on every rank.  per-rank: the one communicator lives on context 0's device state and is
        # used from every context's stream in turn (RCCL orders the collectives of a communicator in issue order)
        comms = [capi.Comm(pipe.contexts[i], uids[i], world, rank) for i in range(n_comms)]
    for s in range(n_slots):
        sl = Slot()
        sl.bgr = make_frames(seed + 1000 * rank + 7919 * s, P, w, h, dev)
        sl.seeds_np = shard.pair_seeds(seed, lo + s * world * P, hi + s * world * P)
        sl.seeds = torch.from_numpy(sl.seeds_np.view(np.int32)).to(dev)
        sl.out = (Pipeline.alloc_pose_outputs if args.pose else Pipeline.alloc_outputs)(torch, 2 * P, P, K, dev)
        sl.rec = torch.zeros((P, words), dtype=torch.int32, device=dev) if multi else None
        sl.gathered = torch.zeros((world * P, words), dtype=torch.int32, device=dev) if multi else None
        sl.comm = comms[s % n_comms] if n_comms else None
        sl.used = 0
        slots.append(sl)
    rccl_ranks = slots[0].comm.info()[0] if gather_mode == "rccl" else None
    if rccl_ranks is not None and rccl_ranks != world:   # what RCCL itself counts (ncclCommCount), not what the launcher said
        sys.exit(f"bench.py: the communicator holds {rccl_ranks} ranks, WORLD_SIZE is {world}")
    torch.cuda.synchronize(dev)

    def step(k):
        """One pass of the hot path over one batch: handed to the next context of the pipeline (returns once it is queued;
        acquire waits for the batch that used the context n_slots steps ago)."""
        t, c = pipe.acquire()
        sl = slots[t % n_slots]      # the pipeline hands out context ticket % n: a slot's frames, outputs and communicator stay with ITS context
        if args.pose:
            c.frontend_pairs_pose(sl.bgr, P, K, ca, sa, pat, sl.seeds, H, thr, POSE_K(w, h), out=sl.out)
        else:
            c.frontend_pairs(sl.bgr, P, K, ca, sa, pat, sl.seeds, H, thr, out=sl.out)
        if multi:
            # the only exchange on the path: fixed-size per-pair result records to every rank, on this batch's own stream
            c.pack_records(sl.out["F"], sl.out["best"], sl.out["matches"], out=sl.rec)
            if sl.comm is not None:
                sl.comm.gather(c, sl.rec, sl.gathered)
        pipe.commit(t)
        if multi and sl.comm is None:      # rehearsal through host memory (VSLAM_BENCH_GATHER=host)
            pipe.wait(t)
            sl.gathered.copy_(shard.gather_records(sl.rec.cpu(), world, n_items=world * P))
        sl.used += 1

    if args.pmc_calibrate and rank == 0:
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
        b = torch.empty_like(a)
        ctx.debug_stream_copy(a, b, 4)
        ctx.debug_stream_copy(a, b, 16)
        ctx.debug_valu_calib()       # a kernel with the vector pipes 100 % busy: calibrates SQ_ACTIVE_INST_VALU (tools/sq_summary.py)
        ctx.synchronize()
        del a, b

    if args.alone:   # per-kernel event timing on = the library keeps the whole step on the main stream
        for c_ in pipe.contexts:
            c_.prof_enable(True)
    # set-up, not steps: every context allocates its workspaces on its first batch (0.1 s each) and RCCL builds its channels on
    # a communicator's first collective (seconds) -- one untimed pass per context keeps both out of the steps whatever W is
    for k in range(n_slots):
        step(k)
    pipe.drain()
    for k in range(args.warmup):
        step(k)
    pipe.drain()
    for sl in slots:
        sl.used = 0
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    pipe.drain()                 # every batch complete, its gather included (same stream, in front of the batch's event)
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    rank_ms = None
    if multi:
        # the step time is the slowest rank's; every rank's own time travels along so that a scaling curve shows stragglers
        t = torch.tensor([dt], dtype=torch.float64)
        all_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in all_t]
        dt = max(float(x.item()) for x in all_t)

    # sanity on the timed output of every context that took part: each pair produced keypoints, matches and an accepted model
    used = [sl for sl in slots if sl.used > 0]
    for sl in used:
        sl.host = {k: sl.out[k].cpu().numpy() for k in ("best", "n", "F", "matches")}
        # (VSLAM_BENCH_ALLOW_DEGENERATE: tools/ab_kernels.py times kernel variants that produce wrong results on purpose)
        assert os.environ.get("VSLAM_BENCH_ALLOW_DEGENERATE") or (
            (sl.host["best"][:, 0] >= 0).all() and (sl.host["best"][:, 3] >= 8).all() and (sl.host["n"] > (K // 2 if args.data != "photo" else K // 8)).all()), "bench output degenerate"
    gather_ok = None
    if multi:   # what the gather delivered: this rank's block of every used context's last gather is this rank's records
        gather_ok = all(torch.equal(sl.gathered[lo:hi], sl.rec) for sl in used)
        flag = torch.tensor([1 if gather_ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_ok = bool(flag.item())
    host_out = used[0].host
    best = np.concatenate([sl.host["best"] for sl in used])
    n_kp = np.concatenate([sl.host["n"] for sl in used])

    result = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        metric = "frame-pairs/sec (extract+match+RANSAC) @1280×720, 2k kp, 4096 hyp; 1/2/4/8 GPU"
        try:   # BASELINE.json names the metric; use its string verbatim when the file is there
            with open(os.path.join(ROOT, "BASELINE.json")) as fh:
                metric = json.load(fh).get("metric", metric)
        except OSError:
            pass
        data_label = {"easy": "easy data: translated texture + one moving block, 13 % outlier matches",
                      "hard": "SURVEY 8(d) data: rotation + parallax, sub-pixel resampling, 40-45 % outlier matches",
                      "photo": "photographic data: windows of four public-domain photographs, rotation <= 1.5 deg + shift <= 12 px"}[args.data]
        par = f"pairs sharded x{world}; {n_slots} batch{'es' if n_slots > 1 else ''} in flight per GPU (vslam_pipeline_*: {n_slots} context{'s' if n_slots > 1 else ''}, steps round-robin)"
        if multi:
            par += {"rccl": ", vslam_gather_records (RCCL all-gather of result records on each batch's stream)",
                    "host": ", records gathered through host memory over gloo (rehearsal)"}[gather_mode]
        result = {
            "metric": metric,
            "value": world * P * args.steps / dt,
            "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/f32/f64",
            "data": "synthetic (" + data_label + ")",
            "config": {"workload": f"{args.workload}: {w}x{h}, {K} keypoints, {H} hypotheses, batch {P} pairs per GPU, {data_label}",
                       "pairs_per_gpu": P, "batches_in_flight": n_slots, "parallelism": par,
                       **({"pose_stages": "every step is vslam_frontend_pairs_pose (--pose)"} if args.pose else {})},
            "setup_steps": n_slots,   # untimed, before the warm-up: one batch per context (workspace allocation, RCCL channel set-up)
            "mean_keypoints": float(n_kp.mean()), "mean_inlier_matches": float(best[:, 3].mean()),
            "workspace_bytes": pipe.workspace_bytes(),   # the contexts' grow-only workspaces for this batch shape (inputs / outputs not counted)
            "workspace_bytes_per_context": ctx.workspace_bytes(),
        }
        if rank_ms:
            result["per_rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "ranks": rank_ms}
        if multi:
            result["record_gather"] = {"through": gather_mode, "rccl_ranks": rccl_ranks, "own_block_intact_on_every_rank": gather_ok,
                                       "words_per_rank": P * words, "communicators_per_rank": n_comms, "comm": args.comm if n_comms else None}
        if args.solver != "exact":
            result["solver"] = "gram-mfma: opt-in approximate 8-point solver, results NOT bit-exact with the reference path"
            result["metric"] += " [NON-PARITY SOLVER]"

    exit_code = 0
    if multi and not gather_ok:
        exit_code = 4

    def timed_single(c, frames, pairs, kk, hh, sd, steps=10, out=None):
        """ms per step of one batch after the other on ONE context."""
        o = c.frontend_pairs(frames, pairs, kk, ca, sa, pat, sd, hh, thr, out=out)
        for _ in range(2):
            o = c.frontend_pairs(frames, pairs, kk, ca, sa, pat, sd, hh, thr, out=o)
        c.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            o = c.frontend_pairs(frames, pairs, kk, ca, sa, pat, sd, hh, thr, out=o)
        c.synchronize()
        return (time.perf_counter() - t1) / steps * 1e3, o

    def profile_pass(c, frames, pairs, kk, hh, sd, psteps):
        """Per-kernel durations with HIP events on the kernels' own stream; one context, one batch after the other, so that a
        kernel's time is its own (with batches in flight the kernels of different batches share the chip)."""
        c.prof_enable(True)
        c.prof_reset()
        for _ in range(psteps):
            c.frontend_pairs(frames, pairs, kk, ca, sa, pat, sd, hh, thr)
        rep = c.prof_report()
        c.prof_enable(False)
        return rep

    def kernel_table(rep, psteps, ww, hh_, kk, hyp_, m_prelim, pairs):
        gray_fused = "bgr2gray_kernel" not in rep
        ks = []
        for name, (ms, cnt) in rep.items():
            per_launch_ms = ms / max(cnt, 1)
            alg = algorithmic_bytes(name, ww, hh_, kk, hyp_, m_prelim, gray_fused) * pairs
            ks.append({"kernel": name, "ms_per_launch": per_launch_ms, "launches_per_step": cnt / psteps,
                       "alg_bytes_per_launch": alg,
                       "alg_GBps": alg / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0})
        ks.sort(key=lambda k: -k["ms_per_launch"] * k["launches_per_step"])
        return ks

    def units_of(kname, ww, hh_, hyp_, pairs, m_prelim, mean_kp):
        """units of work one launch processes (M = inlier matches, a lower bound of the evaluated ones)"""
        if kname == "min_eigen_kernel":
            return 2.0 * ww * hh_ * pairs
        if kname == "ransac_solve_kernel":
            return float(hyp_) * pairs
        return (hyp_ * m_prelim if kname.startswith("ransac") else mean_kp ** 2) * pairs

    def match_view(mk, units, full_batch):
        mv = arithmetic_view("match_knn2_kernel", units, mk["ms_per_launch"], full_batch)
        return {"kernel": "match_knn2_kernel", "bound": "mfma", "achieved": mv["achieved"], "peak": mv["peak"],
                "unit": "TOP/s (FP4, dense)", "frac": mv["frac"],
                "frac_of_int8_peak": mv["achieved"] / INT8_MFMA_PEAK_TOPS,   # the form rounds 1-3 used
                "traffic": pmc_traffic("match_knn2_kernel") if full_batch else None,
                "avg_launch_ms": mk["ms_per_launch"],
                "hbm": {"achieved": mk["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mk["alg_GBps"] / HBM_PEAK_GBS},
                "arithmetic": mv}

    # ---- separate pass: per-kernel durations with HIP events on the kernels' own stream (context 0, its own batch)
    if rank == 0 and not args.no_profile_pass:
        s0 = slots[0]
        psteps = max(1, min(3, args.steps))
        rep = profile_pass(ctx, s0.bgr, P, K, H, s0.seeds, psteps)
        M = float(best[:, 3].mean())
        m_prelim = M   # inlier matches; preliminary matches are >= this
        kernels = kernel_table(rep, psteps, w, h, K, H, m_prelim, P)
        top = kernels[0]
        full_batch = args.workload == "C3" and P == WORKLOADS["C3"][4] and not args.width   # the shape the committed counter passes ran
        if full_batch:   # every kernel of the step against the vector-pipe and HBM ceilings, where this build's counter rows exist
            for k in kernels:
                sq = sq_counters(k["kernel"])
                t_ = k["ms_per_launch"] * 1e-3
                if sq and t_ > 0:
                    k["valu_issue_frac"] = sq[0] * sq[1] / t_ / 1e9 / VALU_PEAK_GINST
                prof = sq_profiled(k["kernel"])
                if prof:
                    k["occupancy_waves_per_simd"] = prof["occupancy_waves_per_simd"]
                    if "valu_busy_frac" in prof:
                        k["valu_busy_frac"] = prof["valu_busy_frac"]
                        k["valu_busy_calibrated"] = prof["valu_busy_calibrated"]
                tr = pmc_traffic(k["kernel"])
                if tr is not None:
                    k["hbm_traffic_bytes_per_launch"] = tr
                    k["hbm_traffic_frac_of_peak"] = tr / t_ / 1e9 / HBM_PEAK_GBS if t_ > 0 else None
        traffic = pmc_traffic(top["kernel"]) if full_batch else None
        hbm_view = {"achieved": top["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top["alg_GBps"] / HBM_PEAK_GBS,
                    "alg_bytes_per_launch": top["alg_bytes_per_launch"]}
        mean_kp = float(n_kp.mean())
        by_name = {k["kernel"]: k for k in kernels}
        av = arithmetic_view(top["kernel"], units_of(top["kernel"], w, h, H, P, m_prelim, mean_kp), top["ms_per_launch"], full_batch)
        # The dominant kernel (ransac_solve or min_eigen) is held by the vector pipe, not by bandwidth.  Two
        # fractions at the top level: `frac` = the vector-instruction issue rate -- waves x instructions per wave (committed
        # SQ counters of this build) / this run's launch time, against the chip's 1228.8 G wave-instructions/s: how busy the
        # binding unit is -- and `frac_algorithmic` = SURVEY 8(d)'s quantity: algorithmic flops per unit x units per launch /
        # launch time against the FP32 vector peak: how much useful work per second.  The HBM view is beside them.
        flops = {k: av[k] for k in ("unit_of_work", "units_per_launch", "ops_per_unit", "op_kind", "achieved", "peak", "unit", "frac") if k in av} if av else None
        if av and "valu_issue" in av:
            vi = av["valu_issue"]
            result["roofline"] = {"kernel": top["kernel"], "bound": "valu", "achieved": vi["achieved"], "peak": vi["peak"],
                                  "unit": vi["unit"], "frac": vi["frac"],
                                  "frac_algorithmic": flops["frac"] if flops else None,
                                  "traffic": traffic, "avg_launch_ms": top["ms_per_launch"],
                                  "how": f"{vi['waves_per_launch']:.0f} waves x {vi['valu_insts_per_wave']:.0f} VALU instructions per wave "
                                         f"({vi['source']}) / {top['ms_per_launch']:.4f} ms / 1e6 = achieved; peak = 1024 SIMDs x 2.4 GHz / 2 "
                                         "cycles per wave instruction; frac_algorithmic = flops.frac (counted flop per hypothesis x "
                                         "hypotheses per launch / launch time / 157.3 TFLOP/s)",
                                  "valu_busy": av.get("valu_busy"), "occupancy_waves_per_simd": av.get("occupancy_waves_per_simd"),
                                  "hbm": hbm_view, "flops": flops or None}
        else:
            result["roofline"] = {"kernel": top["kernel"], "bound": "hbm", "achieved": top["alg_GBps"], "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": top["alg_GBps"] / HBM_PEAK_GBS,
                                  "frac_algorithmic": flops["frac"] if flops else None,
                                  "traffic": traffic, "avg_launch_ms": top["ms_per_launch"],
                                  "note": "no counter rows of this build and shape under profiles/: only the bandwidth view and the "
                                          "algorithmic flop fraction can be formed; the kernel is bound by its vector arithmetic (DESIGN.md 5)"}
            if av:
                result["roofline"]["arithmetic"] = av
        result["roofline"]["measured"] = ("per-kernel times: HIP events on the kernels' own stream, one context, one batch after the other "
                                          "(with batches in flight the kernels of different batches share the chip)")
        if "match_knn2_kernel" in by_name:   # north_star names the match kernel: always report it
            result["roofline_match"] = match_view(by_name["match_knn2_kernel"],
                                                  units_of("match_knn2_kernel", w, h, H, P, m_prelim, mean_kp), full_batch)
        # the HBM-class (stencil) kernel that moves the most bytes, for the bandwidth view of the step
        stencil = [k for k in kernels if k["kernel"] in ("min_eigen_kernel", "gaussian7_kernel", "bgr2gray_kernel")]
        if stencil:
            st = max(stencil, key=lambda k: k["alg_bytes_per_launch"])
            result["roofline_stencil"] = {"kernel": st["kernel"], "bound": "hbm", "achieved": st["alg_GBps"],
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": st["alg_GBps"] / HBM_PEAK_GBS,
                                          "traffic": pmc_traffic(st["kernel"]) if full_batch else None,
                                          "avg_launch_ms": st["ms_per_launch"]}
        result["counter_profiles"] = {"tag": PROFILE_TAG, "match_this_build": counters_current(),
                                      "dominant_kernel_rows_match": counters_current(top["kernel"]),
                                      "occupancy": "kernels[*].occupancy_waves_per_simd = mean resident waves per SIMD while the kernel runs (8 = the "
                                                   "hardware's limit), from SQ_WAVE_CYCLES and GRBM_GUI_ACTIVE of the committed pass",
                                      "note": "traffic / valu_issue / valu_busy come from the committed rocprofv3 PMC summaries and are "
                                              "omitted (null) for a kernel whose source file, or the shared headers, changed since"}
        result["kernels"] = kernels
        result["profile_pass_ms_per_step"] = sum(k["ms_per_launch"] * k["launches_per_step"] for k in kernels)
        busy_ms = sum(k["valu_busy_frac"] * k["ms_per_launch"] * k["launches_per_step"] for k in kernels if "valu_busy_frac" in k)
        if busy_ms > 0:
            # The step against the ceiling that binds it: every kernel's launch time weighted by the share of the chip's vector-pipe
            # time it fills (calibrated, from the committed SQ pass) = the time the step's vector instructions need on a chip whose
            # pipes never idle; divided by the step's wall time with batches in flight.  The path is arithmetic-bound:
            # this, not an HBM fraction, says how close the whole step runs to what its instruction stream allows.
            result["step_vector_pipe"] = {"busy_ms_per_step": busy_ms, "ms_per_step": ms_step, "frac": busy_ms / ms_step,
                                          "what": "sum over the step's kernels of (calibrated vector-pipe share x launch time) / wall time "
                                                  "per step with the batches in flight"}

    # ---- the oracle on the timed batches' own bytes: parity of what was timed -- a share from EVERY context -- and the CPU baseline
    if rank == 0:
        n_cpu = min(P * len(used), args.cpu_pairs if (world == 1 and args.cpu_pairs > 0) else 4)
        share = [n_cpu // len(used) + (1 if i < n_cpu % len(used) else 0) for i in range(len(used))]
        pairs_done, secs, bad, per_ctx = 0, 0.0, [], []
        for i, (sl, n_chk) in enumerate(zip(used, share)):
            if n_chk == 0:
                continue
            base, parity = oracle_on_frames(sl.bgr[:n_chk].cpu().numpy(), sl.bgr[P:P + n_chk].cpu().numpy(), sl.seeds_np[:n_chk], K, H, thr,
                                            sl.host, 0, args.workload + ", " + args.data + " data")
            pairs_done += n_chk
            secs += n_chk / base["value"]
            per_ctx.append({"context": slots.index(sl), "pairs": n_chk, "bit_exact": parity["bit_exact"]})
            bad += [(slots.index(sl), g) for g in parity.get("mismatching_pairs", [])]
        result["parity_in_bench"] = {"pairs": pairs_done, "bit_exact": not bad, "per_context": per_ctx,
                                     "checked": "keypoint counts, inlier-match lists and F (as uint32) of the timed steps' output vs the oracle "
                                                "on the same bytes, a share of the pairs from the last batch of every context"}
        if bad:
            result["parity_in_bench"]["mismatching_context_pair"] = bad[:16]
            exit_code = 3
        if world == 1 and args.cpu_pairs > 0:
            result["cpu_baseline"] = {"value": pairs_done / secs, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                      "sample": f"{pairs_done} pairs of the timed batches themselves ({args.workload}, {args.data} data: {w}x{h}, {K} kp, "
                                                f"{H} hyp) copied back from the device, oracle single thread, {secs:.1f} s"}
            if args.cpu_all_cores_pairs > 0:
                result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.workload, args.cpu_all_cores_pairs, seed)

    # ---- secondary measurements (rank 0, one GPU): one context, the in-flight sweep, the other data regime, the
    # full-evaluation worst case, the other workloads with their own per-kernel pass
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            s0 = slots[0]

            def child(extra, timeout=600):
                """Another configuration of this bench in a FRESH PROCESS (child, never exec): which hardware queue a stream lands on
                depends on what a process created before it, and that mapping moves a step by several per cent -- a second pipeline made
                in this process would not be comparable with the first (HISTORY.md, round 5).  Returns the child's line; a child that
                times out, exits non-zero or prints no line comes back as {"failed": True, ...} with the tail of its stderr (a parity
                mismatch -- exit code 3 -- still carries its line under "line")."""
                import subprocess
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extras", "--cpu-pairs", "0", "--cpu-all-cores-pairs", "0",
                       "--data", args.data, "--solver", args.solver] + extra
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VSLAM_BENCH_FORCE_DIST")}
                try:
                    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)
                except subprocess.TimeoutExpired as e:
                    err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
                    return {"failed": True, "reason": f"no result within {timeout} s", "args": extra, "stderr_tail": err[-600:]}
                except OSError as e:
                    return {"failed": True, "reason": repr(e), "args": extra}
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                line = None
                if lines:
                    try:
                        line = json.loads(lines[-1])
                    except ValueError:
                        line = None
                if r.returncode != 0 or line is None:
                    return {"failed": True, "returncode": r.returncode, "args": extra, "stderr_tail": r.stderr[-600:], "line": line}
                return line

            def child_parity_ok(c):
                """False when a child's own in-bench parity check found a difference (its line is kept; the exit code says so)."""
                line = c.get("line") if c.get("failed") else c
                return not (line and line.get("parity_in_bench") and line["parity_in_bench"].get("bit_exact") is False)

            sweep = {}
            for d in (1, 2, 3, 4, 6):
                if d == n_slots and args.steps >= 20:
                    sweep[str(d)] = {"contexts": d, "ms_per_batch": ms_step, "frame_pairs_per_s": P / ms_step * 1e3, "from": "the timed loop above"}
                    continue
                c = child(["--workload", args.workload, "--pairs", str(P), "--in-flight", str(d), "--steps", "40", "--warmup", "8", "--no-profile-pass"])
                if not child_parity_ok(c):
                    exit_code = 3
                if c.get("failed"):
                    sweep[str(d)] = {"contexts": d, **{k: v for k, v in c.items() if k != "line"}}
                else:
                    sweep[str(d)] = {"contexts": d, "ms_per_batch": c["ms_per_step"], "frame_pairs_per_s": c["value"],
                                     "parity_in_bench": c["parity_in_bench"]["bit_exact"], "from": "a fresh process, 40 steps"}
            result["in_flight_sweep"] = sweep
            if "1" in sweep and not sweep["1"].get("failed"):
                result["single_context"] = {"ms_per_step": sweep["1"]["ms_per_batch"], "frame_pairs_per_s": sweep["1"]["frame_pairs_per_s"],
                                            "what": "one batch after the other on one context (the headline arrangement of rounds 1-4), same data regime, fresh process"}

            def kernel_ms(c, frames, pairs, kk, hh, sd, names):
                rep = profile_pass(c, frames, pairs, kk, hh, sd, 1)
                return {nm: rep[nm][0] / max(rep[nm][1], 1) for nm in names if nm in rep}

            scoring = ("ransac_rank_kernel", "ransac_screen_kernel", "ransac_cand_kernel", "ransac_count_kernel",
                       "ransac_ties_kernel", "ransac_tiesum_kernel", "ransac_select_kernel", "ransac_score_kernel")
            regimes = {}
            for kind in [args.data] + [k_ for k_ in ("hard", "easy", "photo") if k_ != args.data]:
                frames = s0.bgr if kind == args.data else FRAME_MAKERS[kind](seed, P, w, h, dev)
                ms, o = timed_single(ctx, frames, P, K, H, s0.seeds)
                ho = {k: o[k].cpu().numpy() for k in ("best", "n", "F", "matches")}
                cs = ctx.corner_stats()   # of the batch just run: what only the data decides
                entry = {"ms_per_step": ms, "frame_pairs_per_s": P / ms * 1e3, "contexts": 1, "mean_keypoints": float(ho["n"].mean()),
                         "mean_inlier_matches": float(ho["best"][:, 3].mean()),
                         "corner_detector": {"share_of_pixels_sent_to_the_exact_tier": cs["listed_pixels"] / float(cs["frames"] * cs["px_per_frame"]),
                                             "frames_whose_bounded_list_overflowed_per_batch": cs["frames_overflowed"],
                                             "whole_image_sets_in_the_pool": cs["pool_sets"], "frames_per_batch": cs["frames"]},
                         "scoring_kernels_ms": kernel_ms(ctx, frames, P, K, H, s0.seeds, scoring),
                         "front_kernels_ms": kernel_ms(ctx, frames, P, K, H, s0.seeds,
                                                       ("min_eigen_kernel", "corner_exact_kernel", "corner_select_kernel", "match_knn2_kernel",
                                                        "ransac_solve_kernel"))}
                if kind != args.data:   # this regime with batches in flight too, in a process of its own (with its own parity check)
                    cf = child(["--workload", args.workload, "--pairs", str(P), "--data", kind, "--in-flight", str(n_slots), "--steps", "24",
                                "--warmup", "4", "--no-profile-pass", "--cpu-pairs", "8"])
                    if not child_parity_ok(cf):
                        exit_code = 3
                    entry["in_flight"] = ({k: v for k, v in cf.items() if k != "line"} if cf.get("failed") else
                                          {"contexts": n_slots, "ms_per_step": cf["ms_per_step"], "frame_pairs_per_s": cf["value"],
                                           "parity_in_bench": cf["parity_in_bench"]["bit_exact"], "from": "a fresh process, 24 steps"})
                # worst case of the data-dependent scoring kernels: every (hypothesis, match) pair evaluated, every sum formed
                ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, True)
                ms_all, _ = timed_single(ctx, frames, P, K, H, s0.seeds, steps=5)
                entry["full_evaluation"] = {"ms_per_step": ms_all, "scoring_kernels_ms": kernel_ms(ctx, frames, P, K, H, s0.seeds, scoring),
                                            "what": "VSLAM_OPT_RANSAC_ALL_SUMS: no bail-out, no screen: the count and residual sum of every hypothesis"}
                ctx.set_option(ctx.OPT_RANSAC_ALL_SUMS, False)
                if kind != args.data:   # the timed batches were checked above; check this regime's output too
                    nchk = min(P, 24)
                    _, par = oracle_on_frames(frames[:nchk].cpu().numpy(), frames[P:P + nchk].cpu().numpy(), s0.seeds_np[:nchk], K, H, thr, ho, 0,
                                              args.workload + ", " + kind + " data")
                    entry["parity_in_bench"] = par
                    if not par["bit_exact"]:
                        exit_code = 3
                    del frames
                regimes[kind] = entry
            result["data_regimes"] = regimes

            # ---- the SURVEY 8(f) rows at batch scale: the resident pose chain (vslam_frontend_pairs_pose = the step + extract_Rt +
            # triangulate + reprojection filter, src/vslam.cpp:82-88,120-125,186-251) and the map-association block
            # (src/vslam.cpp:129-161) on the batch's own results, one context; checked against the oracle on a few pairs
            def pose_and_association():
                Kmat = POSE_K(w, h)
                po = ctx.frontend_pairs_pose(s0.bgr, P, K, ca, sa, pat, s0.seeds, H, thr, Kmat)
                for _ in range(2):
                    ctx.frontend_pairs_pose(s0.bgr, P, K, ca, sa, pat, s0.seeds, H, thr, Kmat, out=po)
                ctx.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    ctx.frontend_pairs_pose(s0.bgr, P, K, ca, sa, pat, s0.seeds, H, thr, Kmat, out=po)
                ctx.synchronize()
                ms_pose = (time.perf_counter() - t1) / 10 * 1e3
                ms_plain, _ = timed_single(ctx, s0.bgr, P, K, H, s0.seeds)
                ctx.prof_enable(True)
                ctx.prof_reset()
                ctx.frontend_pairs_pose(s0.bgr, P, K, ca, sa, pat, s0.seeds, H, thr, Kmat, out=po)
                rep2 = ctx.prof_report()
                ctx.prof_enable(False)
                stage = {nm: rep2[nm][0] / max(rep2[nm][1], 1) for nm in ("pose_from_F_kernel", "triangulate_kernel", "reproj_filter_kernel") if nm in rep2}
                hp = {k_: v_.cpu().numpy() for k_, v_ in po.items()}
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from oracle_lib import Oracle
                o_ = Oracle()
                bits = lambda a_: np.ascontiguousarray(a_, np.float32).view(np.uint32)
                c1 = np.c_[Kmat, np.zeros(3, np.float32)]
                bad_p = []
                for b_ in range(min(P, 8)):
                    Rr, tr_ = o_.extract_Rt(hp["F"][b_], Kmat)
                    c2r = o_.camera_matrix(Kmat, Rr, tr_)
                    kk = int(hp["best"][b_, 3])
                    mm = hp["matches"][b_, :kk]
                    p1_, p2_ = hp["xy"][b_][mm[:, 0]], hp["xy"][P + b_][mm[:, 1]]
                    ref = o_.triangulate(p1_, p2_, c1, c2r)
                    kept, err = o_.reprojection_filter(ref, p1_, p2_, c1, c2r, np.full(kk, -1, np.int32), 4.0)
                    ok_ = (np.array_equal(bits(hp["R"][b_]), bits(Rr.reshape(9))) and np.array_equal(bits(hp["t"][b_]), bits(tr_))
                           and np.array_equal(bits(hp["c2"][b_]), bits(c2r.reshape(12))) and np.array_equal(bits(hp["points4d"][b_, :kk]), bits(ref))
                           and int(hp["n_inliers"][b_]) == len(kept) and np.array_equal(hp["inlier_idx"][b_, :len(kept)], kept)
                           and float(hp["error"][b_]) == err)
                    if not ok_:
                        bad_p.append(b_)
                entry = {"what": "vslam_frontend_pairs_pose on the timed batch, one context: the step + extract_Rt + triangulate + reprojection filter, nothing leaving the device",
                         "ms_per_step": ms_pose, "ms_per_step_without_the_pose_stages": ms_plain, "pose_stages_ms": ms_pose - ms_plain,
                         "kernels_ms_per_launch": stage, "mean_reprojection_inliers": float(hp["n_inliers"].mean()),
                         "parity_in_bench": {"pairs": min(P, 8), "bit_exact": not bad_p,
                                             "checked": "R, t, c2, the triangulated points (as uint32), the reprojection filter's index list and error sum vs the oracle"}}
                # association: every pair's triangulated points as the map, one observation each (the matched keypoint of the FIRST
                # frame), searched for in the SECOND frame's tree with the loop's radius 2 and threshold 64
                Mp = K
                mp_ = po["points4d"]                                     # (P, K, 4), rows beyond a pair's inliers are zero
                n_map = po["best"][:, 3].contiguous().to(torch.int32)
                offs = torch.arange(Mp + 1, dtype=torch.int32, device=dev).repeat(P, 1).contiguous()   # one observation per map point
                idx1 = po["matches"][:, :, 0].long().clamp(0, K - 1)
                od = torch.gather(po["desc"][:P], 1, idx1[:, :, None].expand(P, K, 32)).contiguous()
                ids_all = torch.full((12, P, K), -1, dtype=torch.int32, device=dev)   # map_point_ids is in / out: a fresh set per call
                claim = torch.full((P, Mp), -3, dtype=torch.int32, device=dev)
                nodes2, xy2, desc2, n2 = po["nodes"][P:].contiguous(), po["xy"][P:].contiguous(), po["desc"][P:].contiguous(), po["n"][P:].contiguous()
                torch.cuda.synchronize(dev)
                ctx.associate(mp_, n_map, po["c2"], w, h, nodes2, xy2, desc2, n2, offs, od, ids_all[11], claim=claim)
                ctx.synchronize()
                t1 = time.perf_counter()
                for i_ in range(10):
                    ctx.associate(mp_, n_map, po["c2"], w, h, nodes2, xy2, desc2, n2, offs, od, ids_all[i_], claim=claim)
                ctx.synchronize()
                ms_assoc = (time.perf_counter() - t1) / 10 * 1e3
                ctx.prof_enable(True)
                ctx.prof_reset()
                ctx.associate(mp_, n_map, po["c2"], w, h, nodes2, xy2, desc2, n2, offs, od, ids_all[10], claim=claim)
                rep3 = ctx.prof_report()
                ctx.prof_enable(False)
                ids_all[9].fill_(-1)
                torch.cuda.synchronize(dev)
                ctx.associate(mp_, n_map, po["c2"], w, h, nodes2, xy2, desc2, n2, offs, od, ids_all[9], claim=claim)   # (the checked run: fresh ids)
                ctx.synchronize()
                hc, hid = claim.cpu().numpy(), ids_all[9].cpu().numpy()
                bad_a = []
                for b_ in range(min(P, 4)):
                    kk, nk = int(hp["best"][b_, 3]), int(hp["n"][P + b_])
                    ref_ids, ref_claim = o_.associate(hp["points4d"][b_, :kk], hp["c2"][b_], w, h, hp["nodes"][P + b_, :nk], hp["xy"][P + b_, :nk],
                                                      hp["desc"][P + b_, :nk], np.arange(kk + 1, dtype=np.int32), od[b_, :max(kk, 1)].cpu().numpy(),
                                                      np.full(nk, -1, np.int32))
                    if not (np.array_equal(hc[b_, :kk], ref_claim) and np.array_equal(hid[b_, :nk], ref_ids)):
                        bad_a.append(b_)
                entry["association"] = {"what": "vslam_associate_map_points: each pair's triangulated inliers as map points (one observation each), radius 2, threshold 64",
                                        "ms_per_batch": ms_assoc, "kernels_ms_per_launch": {nm: rep3[nm][0] / max(rep3[nm][1], 1) for nm in rep3 if rep3[nm][1] > 0},
                                        "map_points_per_pair": float(n_map.float().mean()),
                                        "claimed_per_pair": float((claim >= 0).float().sum(1).mean()),
                                        "parity_in_bench": {"pairs": min(P, 4), "bit_exact": not bad_a}}
                return entry, bool(bad_p or bad_a)
            pc, pc_bad = pose_and_association()
            cp = child(["--workload", args.workload, "--pairs", str(P), "--in-flight", str(n_slots), "--steps", "24", "--warmup", "4",
                        "--no-profile-pass", "--cpu-pairs", "8", "--pose"])
            if not child_parity_ok(cp):
                exit_code = 3
            pc["in_flight"] = ({k: v for k, v in cp.items() if k != "line"} if cp.get("failed") else
                               {"contexts": n_slots, "ms_per_step": cp["ms_per_step"], "frame_pairs_per_s": cp["value"],
                                "parity_in_bench": cp["parity_in_bench"]["bit_exact"],
                                "what": "every step = vslam_frontend_pairs_pose, batches in flight as in the headline; a fresh process, 24 steps"})
            result["pose_chain"] = pc
            if pc_bad:
                exit_code = 3
            others = {}
            for wl in sorted(WORKLOADS):   # each in a process of its own (see child): headline arrangement, own per-kernel pass
                if wl == args.workload:
                    continue
                w2, h2, K2, H2, P2 = WORKLOADS[wl]
                extra_cpu = ["--cpu-pairs", "4"] if wl == "C3g" else []   # (C3g checks its output against the oracle inside the child)
                c = child(["--workload", wl, "--in-flight", str(n_slots), "--steps", "24" if wl != "C5" else "12", "--warmup", "4"] + extra_cpu)
                c1 = child(["--workload", wl, "--in-flight", "1", "--steps", "12" if wl != "C5" else "6", "--warmup", "2", "--no-profile-pass"])
                if not (child_parity_ok(c) and child_parity_ok(c1)):
                    exit_code = 3
                if c1.get("failed"):
                    c1 = None
                if c.get("failed"):
                    others[wl] = {k: v for k, v in c.items() if k != "line"}
                    continue
                kt = c.get("kernels", [])
                entry = {"workload": c["config"]["workload"], "ms_per_step": c["ms_per_step"], "frame_pairs_per_s": c["value"],
                         "batches_in_flight": n_slots, "steps": c["steps"], "parity_in_bench": c["parity_in_bench"],
                         "single_context": {"ms_per_step": c1["ms_per_step"], "frame_pairs_per_s": c1["value"]} if c1 else None,
                         "workspace_bytes_per_context": c["workspace_bytes_per_context"], "mean_inlier_matches": c["mean_inlier_matches"],
                         "kernels_ms_per_launch": {k["kernel"]: round(k["ms_per_launch"], 5) for k in kt}}
                if "roofline" in c:
                    rf = c["roofline"]
                    entry["dominant_kernel"] = {"kernel": rf["kernel"], "ms_per_launch": rf["avg_launch_ms"], "frac_algorithmic": rf.get("frac_algorithmic"),
                                                "hbm_frac_algorithmic": (rf.get("hbm") or {}).get("frac", rf["frac"] if rf["bound"] == "hbm" else None)}
                if "roofline_match" in c:
                    entry["roofline_match"] = c["roofline_match"]
                if "grid_extractor" in c:
                    entry["grid_extractor"] = c["grid_extractor"]
                    entry["cpu_baseline"] = c.get("cpu_baseline")
                others[wl] = entry
            result["other_workloads"] = others
            # a frame width that is no multiple of 4 (padded internal rows under the same kernels): headline arrangement, own parity check
            wodd = w - 2
            cw = child(["--workload", args.workload, "--pairs", str(P), "--width", str(wodd), "--in-flight", str(n_slots), "--steps", "24",
                        "--warmup", "4", "--no-profile-pass", "--cpu-pairs", "8"])
            if not child_parity_ok(cw):
                exit_code = 3
            result["other_shapes"] = {f"{wodd}x{h}": ({k: v for k, v in cw.items() if k != "line"} if cw.get("failed") else
                                                     {"ms_per_step": cw["ms_per_step"], "frame_pairs_per_s": cw["value"], "batches_in_flight": n_slots,
                                                      "x_the_headline_step": cw["ms_per_step"] / ms_step,
                                                      "mean_keypoints": cw["mean_keypoints"], "parity_in_bench": cw["parity_in_bench"],
                                                      "what": "the same workload at a width that is no multiple of 4: cvtColor writes gray rows of a "
                                                              "multiple of 16 bytes with a mirrored tail and the same kernels run on them; a fresh process"})}
        except Exception as e:   # a secondary measurement must not cost the headline its line
            import traceback
            result["extras_error"] = {"error": repr(e), "traceback_tail": traceback.format_exc()[-800:]}
            exit_code = exit_code or 5

    if multi:
        dist.barrier()
        for cm in comms:
            cm.close()
        dist.destroy_process_group()
    pipe.close()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    os.close(real_stdout)
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
