#!/usr/bin/env python3
"""TRAP (round 5): pipelines made one after the other in ONE process do not get the same hardware queues -- three identical
ones ran at 2.917 / 2.669 / 2.749 ms per batch -- so this tool cannot compare anything that touches streams or their order of
creation; use tools/ab_proc.sh (fresh process per configuration) for that.

Context-level knobs with batches in flight, one process, alternating blocks: what helped one batch at a time (forks onto
the auxiliary stream, stream priorities) may not with three batches filling each other's holes.
    python tools/ab_env.py [C3] [in_flight] [rounds] [hard|easy]"""
import os
os.environ.setdefault('VSLAM_AMD_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vslam_amd', 'libvslam_amd_exp.so'))   # the knobs exist in the EXPERIMENTS build only
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vslam_amd import capi, shard, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kind = sys.argv[4] if len(sys.argv) > 4 else "hard"
w, h, K, H, P = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(0x5EED0002, 0, P).view(np.int32)).to(dev)
bgr = (synth.frames_torch_hard if kind == "hard" else synth.frames_torch)(0x5EED0002, P, w, h, dev)
C = capi.Context
NP = {"VSLAM_STREAM_PRIORITY": "0"}
CONFIGS = {
    "default": ({}, {}, True),
    "no priorities, trees in front, blur on main": (dict(NP, VSLAM_OVERLAP_BLUR="0"), {C.OPT_TREE_FORK: 0}, True),
    "no priorities, trees in line, blur on main": (dict(NP, VSLAM_OVERLAP_BLUR="0"), {C.OPT_TREE_FORK: 5}, True),
    "one stream: + generator in line": (dict(NP, VSLAM_OVERLAP_BLUR="0", VSLAM_SETS_PREFETCH="0"), {C.OPT_TREE_FORK: 5}, True),
    "priorities, trees in front, blur on main": (dict(VSLAM_OVERLAP_BLUR="0"), {C.OPT_TREE_FORK: 0}, True),
    "priorities, trees in line, blur on main": (dict(VSLAM_OVERLAP_BLUR="0"), {C.OPT_TREE_FORK: 5}, True),
    "no priorities, trees behind matcher, blur on main": (dict(NP, VSLAM_OVERLAP_BLUR="0"), {C.OPT_TREE_FORK: 1}, True),
    "no priorities, no trees, blur on main": (dict(NP, VSLAM_OVERLAP_BLUR="0"), {}, False),
}
if os.environ.get("AB_SET") == "shared":   # variations around what vslam_pipeline_create chooses by itself (no knob forced)
    CONFIGS = {
        "default": (None, {}, True),
        "matcher 4 x 64": (None, {C.OPT_MATCH_SHAPE: 2}, True),
        "trees behind the matcher": (None, {C.OPT_TREE_FORK: 1}, True),
        "trees in line": (None, {C.OPT_TREE_FORK: 5}, True),
        "generator in line": ({"VSLAM_SETS_PREFETCH": "0"}, {}, True),
        "corner window 120 %": (None, {C.OPT_CORNER_WINDOW_PCT: 120}, True),
        "corner window 160 %": (None, {C.OPT_CORNER_WINDOW_PCT: 160}, True),
        "default again (order check)": (None, {}, True),
    }
if os.environ.get("AB_SET") == "order":   # is the first pipeline made in a process slower than the same made later?
    CONFIGS = {"default": (None, {}, True), "same, made second": (None, {}, True), "same, made third": (None, {}, True)}
if os.environ.get("AB_SET") == "prio":
    CONFIGS = {
        "default": ({}, {}, True),
        "main at default priority, aux low": (dict(VSLAM_STREAM_PRIORITY="2"), {}, True),
        "no priorities": (dict(NP), {}, True),
    }
if os.environ.get("AB_SET") == "small":   # large shapes: fewer configurations (every one holds its own workspaces)
    CONFIGS = {
        "default": ({}, {}, True),
        "no priorities": (dict(NP), {}, True),
        "blur on main": (dict(VSLAM_OVERLAP_BLUR="0"), {}, True),
        "no priorities, blur on main": (dict(NP, VSLAM_OVERLAP_BLUR="0"), {}, True),
        "no priorities, trees behind matcher": (dict(NP), {C.OPT_TREE_FORK: 1}, True),
    }
pipes = {}
for name, (env, opts, _) in CONFIGS.items():
    # explicit values for every knob: "default" here is the ONE-batch-at-a-time arrangement (priorities, blur and k-d build
    # forked), whatever vslam_pipeline_create would choose by itself
    for k_ in ("VSLAM_OVERLAP_BLUR", "VSLAM_STREAM_PRIORITY", "VSLAM_SETS_PREFETCH"):
        os.environ.pop(k_, None)
    if os.environ.get("AB_SET") in ("shared", "order"):
        os.environ.update(env or {})
    else:
        os.environ.update({"VSLAM_OVERLAP_BLUR": "2", "VSLAM_STREAM_PRIORITY": "1", "VSLAM_SETS_PREFETCH": "1"})
        os.environ.update(env)
        opts = {**{C.OPT_TREE_FORK: 1 if K <= 2048 else 0}, **opts}
    pipes[name] = capi.Pipeline(0, depth)
    for o, v in opts.items():
        pipes[name].set_option(o, v)
outs = [capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev) for _ in range(depth)]
outs_nt = [dict(o, nodes=None) for o in outs]
steps = 30 if wl != "C5" else 9


def run(pp, n, oo):
    for i in range(n):
        t, c = pp.acquire()
        c.frontend_pairs(bgr, P, K, ca, sa, None, seeds, H, 10.0, out=oo[i % depth])
        pp.commit(t)
    pp.drain()


res = {k: [] for k in CONFIGS}
for rnd in range(rounds + 1):
    for name, (_, _, trees) in CONFIGS.items():
        oo = outs if trees else outs_nt
        run(pipes[name], 2 * depth, oo)
        t0 = time.perf_counter()
        run(pipes[name], steps, oo)
        if rnd:
            res[name].append((time.perf_counter() - t0) / steps * 1e3)
base = np.median(res["default"])
for name, v in res.items():
    print(f"{wl} {kind} {depth} in flight  {name:50s} median {np.median(v):.4f} ms  min {np.min(v):.4f}  (+-{np.std(v):.4f})  {np.median(v) - base:+.4f}", flush=True)
