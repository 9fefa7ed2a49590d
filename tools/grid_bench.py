"""Time the grid ORB/FAST extractor (extract_features(Frame&, nrows, ncols), src/Frame.cpp:16-51) alone:
F frames of w x h through vslam_extract_features_grid, per-scope HIP-event times, ms per frame.
  python tools/grid_bench.py [--frames 64] [--w 1280 --h 720] [--grid 4] [--steps 10] [--data hard|easy|photo|noise]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--w", type=int, default=1280)
    ap.add_argument("--h", type=int, default=720)
    ap.add_argument("--grid", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cap", type=int, default=8192)
    ap.add_argument("--data", default="hard")
    ap.add_argument("--no-prof", action="store_true")
    a = ap.parse_args()
    import torch
    from vslam_amd import Context, synth
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    P = a.frames // 2
    if a.data == "noise":
        bgr = torch.randint(0, 256, (a.frames, a.h, a.w, 3), dtype=torch.uint8, device=dev)
    elif a.data == "photo":   # windows of the four photographs of tests/golden/real_v1.npz under small camera motions
        bgr = synth.frames_torch_photo(0x5EED0002, P, a.w, a.h, dev)
    else:
        mk = synth.frames_torch if a.data == "easy" else synth.frames_torch_hard
        bgr = mk(0x5EED0002, P, a.w, a.h, dev)
    pat = torch.from_numpy(synth.brief_pattern()).to(dev)
    work = bgr.clone()
    out = ctx.extract_features_grid(work, a.grid, a.grid, pat, a.cap)
    ctx.synchronize()
    n = out["n"].cpu().numpy()
    for _ in range(2):
        ctx.extract_features_grid(work, a.grid, a.grid, pat, a.cap)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.extract_features_grid(work, a.grid, a.grid, pat, a.cap)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    res = {"frames": a.frames, "w": a.w, "h": a.h, "grid": a.grid, "data": a.data, "ms_per_call": ms,
           "us_per_frame": ms / a.frames * 1e3, "mean_keypoints": float(n.mean()), "max_keypoints": int(n.max()),
           "workspace_bytes": ctx.workspace_bytes()}
    if not a.no_prof:
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(3):
            ctx.extract_features_grid(work, a.grid, a.grid, pat, a.cap)
        rep = ctx.prof_report()
        ctx.prof_enable(False)
        res["scopes_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in rep.items()}
    print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
