#!/usr/bin/env python3
"""The documented performance cliffs, timed: layouts and content that leave the fast paths.
One context, one batch after the other, 64 pairs of about 1280 x 720, 2000 keypoints, 4096 hypotheses.
  * width not a multiple of 4: until round 6 the plain whole-image corner pipeline and the per-keypoint descriptor kernel,
    now padded internal rows (vslam_ctx::img_pitch) under the same kernels;
  * rows not dword-aligned (row stride 3 w + 1): until round 6 bgr2gray as a launch of its own in front of the gray-input
    detector, now the fused detector on unaligned loads;
  * noise frames: every frame overflows the bounded corner lists; up to the pool's sets are redone from whole-image scratch,
    beyond that the call fails with VSLAM_ERR_CAPACITY and has to be repeated with VSLAM_OPT_CORNER_LIST_CAP = -1.
    python tools/cliffs.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, VslamError, capi, shard, synth  # noqa: E402

P, K, H = 64, 2000, 4096
dev = torch.device("cuda", 0)
ctx = Context(0)
ca, sa = synth.keypoint_rotation()
seeds = torch.from_numpy(shard.pair_seeds(7, 0, P).view(np.int32)).to(dev)


def run(bgr, w, h, stride, steps=8):
    """frontend_pairs on rows of `stride` bytes (bgr: (2 P, h, stride) uint8)."""
    out = capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev)
    p = ctx._params(K, ca, sa, None)

    def once():
        ctx._check(ctx.lib.vslam_frontend_pairs(ctx.handle, C.c_void_p(bgr.data_ptr()), C.c_int(P), C.c_int(w), C.c_int(h), C.c_int(stride),
                                                C.byref(p), C.c_int(K), C.c_void_p(seeds.data_ptr()), C.c_int(H), C.c_float(10.0),
                                                C.c_void_p(out["xy"].data_ptr()), C.c_void_p(out["desc"].data_ptr()),
                                                C.c_void_p(out["nodes"].data_ptr()), C.c_void_p(out["n"].data_ptr()),
                                                C.c_void_p(out["matches"].data_ptr()), C.c_void_p(out["best"].data_ptr()),
                                                C.c_void_p(out["F"].data_ptr())))
    for _ in range(2):
        once()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    ctx.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(out["n"].float().mean()), float(out["best"][:, 3].float().mean())


def padded(frames, stride):
    f, h, w, _ = frames.shape
    buf = torch.zeros((f, h, stride), dtype=torch.uint8, device=dev)
    buf[:, :, :3 * w] = frames.reshape(f, h, 3 * w)
    return buf


base = synth.frames_torch_hard(11, P, 1280, 720, dev)
rows = []
ms, nk, nm = run(base.reshape(2 * P, 720, 3 * 1280), 1280, 720, 3 * 1280)
rows.append(("1280 x 720, packed rows (the fast path)", ms, nk, nm))
ms, nk, nm = run(padded(base, 3 * 1280 + 4), 1280, 720, 3 * 1280 + 4)
rows.append(("1280 x 720, row stride 3 w + 4 (dword-aligned padding)", ms, nk, nm))
ms, nk, nm = run(padded(base, 3 * 1280 + 1), 1280, 720, 3 * 1280 + 1)
rows.append(("1280 x 720, row stride 3 w + 1 (rows not dword-aligned)", ms, nk, nm))
narrow = base[:, :, :1278].contiguous()
ms, nk, nm = run(narrow.reshape(2 * P, 720, 3 * 1278), 1278, 720, 3 * 1278)
rows.append(("1278 x 720 (width not a multiple of 4)", ms, nk, nm))
g = torch.Generator(device=dev).manual_seed(5)
noise = torch.randint(0, 256, (2 * P, 720, 1280, 3), dtype=torch.uint8, device=dev, generator=g)
try:
    ms, nk, nm = run(noise.reshape(2 * P, 720, 3 * 1280), 1280, 720, 3 * 1280, steps=3)
    rows.append(("1280 x 720 noise (lists overflow; pool large enough)", ms, nk, nm))
except VslamError as e:
    rows.append((f"1280 x 720 noise: {str(e)[:60]}...", float("nan"), 0, 0))
# the same noise batch through vslam_pipeline_submit_pairs: the pipeline queues the overflowing batch a second time with
# whole-image lists when its status is collected (round 6), so the ticket reports VSLAM_OK; the time is per batch, the
# failed first attempt included
pipe = capi.Pipeline(0, 2)
try:
    outs = [capi.Pipeline.alloc_outputs(torch, 2 * P, P, K, dev) for _ in range(2)]
    torch.cuda.synchronize()
    nz = noise.reshape(2 * P, 720, 1280, 3)
    for o_ in outs:
        pipe.submit_pairs(nz, P, K, ca, sa, None, seeds, H, 10.0, o_)
    pipe.drain()
    r0 = pipe.batches_redone()
    t0 = time.perf_counter()
    for i in range(4):
        pipe.submit_pairs(nz, P, K, ca, sa, None, seeds, H, 10.0, outs[i % 2])
    pipe.drain()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    rows.append((f"1280 x 720 noise through vslam_pipeline_submit_pairs ({pipe.batches_redone() - r0} of 4 batches done again: VSLAM_OK)",
                 ms, float(outs[0]["n"].float().mean()), float(outs[0]["best"][:, 3].float().mean())))
except VslamError as e:
    rows.append((f"1280 x 720 noise through the pipeline: {str(e)[:60]}...", float("nan"), 0, 0))
finally:
    pipe.close()
ctx.set_option(ctx.OPT_CORNER_LIST_CAP, -1)
ms, nk, nm = run(noise.reshape(2 * P, 720, 3 * 1280), 1280, 720, 3 * 1280, steps=3)
rows.append(("1280 x 720 noise, VSLAM_OPT_CORNER_LIST_CAP = -1 (whole-image lists)", ms, nk, nm))
ms, nk, nm = run(base.reshape(2 * P, 720, 3 * 1280), 1280, 720, 3 * 1280)
rows.append(("1280 x 720 image data, VSLAM_OPT_CORNER_LIST_CAP = -1", ms, nk, nm))
ctx.set_option(ctx.OPT_CORNER_LIST_CAP, 0)
odd = base[:, :, :1277].contiguous()
ms, nk, nm = run(padded(odd, 3 * 1277 + 2), 1277, 720, 3 * 1277 + 2)
rows.append(("1277 x 720, row stride 3 w + 2 (odd width AND unaligned rows)", ms, nk, nm))
# the grid ORB/FAST extractor (src/Frame.cpp:16-51) alone, 64 frames, 4 x 4 cells: level 0 of its pyramids is the gray frame
grid_rows = []
pat = torch.from_numpy(synth.brief_pattern()).to(dev)
for wg in (1280, 1278):
    fr = base[:64, :, :wg].contiguous()
    o_ = ctx.extract_features_grid(fr, 4, 4, pat, 8192)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        ctx.extract_features_grid(fr, 4, 4, pat, 8192, out=o_)
    ctx.synchronize()
    grid_rows.append((wg, (time.perf_counter() - t0) / 8 * 1e3, float(o_["n"].float().mean())))
print(f"{P} pairs, {K} keypoints, {H} hypotheses, one context; ms per batch, x the fast path, keypoints / inlier matches per frame / pair")
for name, ms, nk, nm in rows:
    print(f"  {name:72s} {ms:8.3f}  x{ms / rows[0][1]:5.2f}   {nk:7.1f} {nm:7.1f}")
print("grid ORB/FAST extractor alone, 64 frames, 4 x 4 cells; ms per call, x the width-1280 call, keypoints per frame")
for wg, ms, nk in grid_rows:
    print(f"  {wg} x 720{'' if wg % 4 == 0 else ' (width not a multiple of 4)':40s} {ms:8.3f}  x{ms / grid_rows[0][1]:5.2f}   {nk:7.1f}")
