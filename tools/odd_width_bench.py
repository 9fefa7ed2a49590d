"""The front-end step at a width that is no multiple of 4 (padded internal rows, vslam_ctx::img_pitch) beside the same frames cut to
a multiple of 4: ms per batch and per-stage event times -- the same kernels run in both (tools/_odd_width_prof.sh puts rocprofv3
on it).   python tools/odd_width_bench.py [--pairs 64] [--w 1278] [--h 720]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--w", type=int, default=1278)
    ap.add_argument("--h", type=int, default=720)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--only-odd", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    from vslam_amd import Context, shard, synth
    P, K, H, thr = a.pairs, 2000, 4096, 10.0
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ca, sa = synth.keypoint_rotation()
    seeds = torch.from_numpy(shard.pair_seeds(0x5EED0001, 0, P).view(np.int32)).to(dev)
    wide = (a.w + 3) & ~3
    full = synth.frames_torch_hard(0x5EED0001, P, wide, a.h, dev)
    res = {}
    for w in ([a.w] if a.only_odd else [wide, a.w]):
        fr = full[:, :, :w, :].contiguous()
        o = ctx.frontend_pairs(fr, P, K, ca, sa, None, seeds, H, thr)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ctx.frontend_pairs(fr, P, K, ca, sa, None, seeds, H, thr, out=o)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.frontend_pairs(fr, P, K, ca, sa, None, seeds, H, thr, out=o)
        rep = ctx.prof_report()
        ctx.prof_enable(False)
        res[str(w)] = {"ms_per_batch": ms, "mean_keypoints": float(o["n"].float().mean()),
                       "scopes_ms": {k: round(v[0] / v[1], 4) for k, v in rep.items() if v[1] > 0}}
    print(json.dumps({"pairs": P, "height": a.h, "widths": res}))
    ctx.close()


if __name__ == "__main__":
    main()
