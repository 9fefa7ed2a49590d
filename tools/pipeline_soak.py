#!/usr/bin/env python3
"""Soak of vslam_pipeline_*: a queue of batches of uneven size (workspaces regrow between tickets), alternating rBRIEF
tables, N contexts, the pipeline torn down and rebuilt every repetition; every output of every batch compared on the device
with the same batch on a single context.  What a race between the contexts' streams, a stale workspace or a wrong join
would eventually show.      python tools/pipeline_soak.py [contexts] [repetitions] [WxH]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vslam_amd import Context, capi, synth  # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "320x240").split("x"))
MAXC, HYP, THR, SEED = (300, 64, 10.0, 0xFACE) if W <= 320 else (1000, 1024, 10.0, 0xFACE)
SIZES = [2, 1, 3, 1, 4, 2, 1, 5, 1, 2, 3]
ca, sa = synth.keypoint_rotation()
pat = torch.from_numpy(synth.brief_pattern()).cuda()
dev, first = [], 0
for i, n in enumerate(SIZES):
    bgr = torch.from_numpy(synth.frames_numpy(500 + i, n, W, H)).cuda()
    seeds = torch.from_numpy((np.uint32(SEED) ^ np.arange(first, first + n, dtype=np.uint32)).view(np.int32)).cuda()
    dev.append((bgr, seeds))
    first += n
ctx = Context(0)
ref = []
for bgr, seeds in dev:
    n = bgr.shape[0] // 2
    o = ctx.frontend_pairs(bgr, n, MAXC, ca, sa, None, seeds, HYP, THR)
    ctx.synchronize()
    ref.append({k: v.clone() for k, v in o.items()})
    ref[-1]["rec"] = ctx.pack_records(o["F"], o["best"], o["matches"]).clone()
torch.cuda.synchronize()
bad = 0
for rep in range(reps):
    pipe = capi.Pipeline(0, n_ctx)
    outs, recs = [], []
    for b, (bgr, seeds) in enumerate(dev):
        n = bgr.shape[0] // 2
        outs.append(capi.Pipeline.alloc_outputs(torch, 2 * n, n, MAXC, bgr.device))
        recs.append(torch.zeros((n, 13 + MAXC), dtype=torch.int32, device=bgr.device))
        torch.cuda.synchronize()
        pipe.submit_pairs(bgr, n, MAXC, ca, sa, pat if (b + rep) % 2 else None, seeds, HYP, THR, outs[-1], records=recs[-1])
    pipe.wait(3 + rep % 4)
    pipe.drain()
    for b, (o, rc, r) in enumerate(zip(outs, recs, ref)):
        for k in ("n", "best", "F", "xy", "desc", "nodes"):
            if not torch.equal(o[k], r[k]):
                bad += 1
                print(f"rep {rep} batch {b} (context {b % n_ctx}, {SIZES[b]} pairs) differs in {k}", flush=True)
                break
        else:
            if not torch.equal(rc, r["rec"]):
                bad += 1
                print(f"rep {rep} batch {b}: records differ", flush=True)
    pipe.close()
print(f"{n_ctx} contexts, {W}x{H}: {bad} differing batches of {reps * len(SIZES)}")
sys.exit(1 if bad else 0)
