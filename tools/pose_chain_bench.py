"""The SURVEY 8(f) rows at batch scale, alone: vslam_frontend_pairs_pose (the C3 step + extract_Rt + triangulate + the
reprojection filter, src/vslam.cpp:82-88,120-125,186-251, src/helpers.cpp:3-80) and vslam_associate_map_points
(src/vslam.cpp:129-161) on the batch's own triangulated points -- the same calls bench.py times under `pose_chain`, as a
program of their own so that rocprofv3 can sit on them (tools/_pose_prof.sh).
  python tools/pose_chain_bench.py [--pairs 256] [--steps 10] [--no-prof]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--pmc-calibrate", action="store_true", help="also launch the busy-pipe kernel tools/sq_summary.py calibrates against")
    a = ap.parse_args()
    import numpy as np
    import torch
    from vslam_amd import Context, shard, synth
    import bench
    w, h, K, H, _ = bench.WORKLOADS["C3"]
    P, thr = a.pairs, 10.0
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    seed = 0x5EED0000 + bench.SEED_INDEX["C3"]
    bgr = synth.frames_torch_hard(seed, P, w, h, dev)
    seeds = torch.from_numpy(shard.pair_seeds(seed, 0, P).view(np.int32)).to(dev)
    pat = torch.from_numpy(synth.brief_pattern()).to(dev)
    ca, sa = synth.keypoint_rotation()
    Kmat = np.array([[525.0, 0, w // 2], [0, 525.0, h // 2], [0, 0, 1]], np.float32)   # src/vslam.cpp:32
    po = ctx.frontend_pairs_pose(bgr, P, K, ca, sa, pat, seeds, H, thr, Kmat)
    ctx.synchronize()

    def pose():
        ctx.frontend_pairs_pose(bgr, P, K, ca, sa, pat, seeds, H, thr, Kmat, out=po)

    n_map = po["best"][:, 3].contiguous().to(torch.int32)
    offs = torch.arange(K + 1, dtype=torch.int32, device=dev).repeat(P, 1).contiguous()
    idx1 = po["matches"][:, :, 0].long().clamp(0, K - 1)
    od = torch.gather(po["desc"][:P], 1, idx1[:, :, None].expand(P, K, 32)).contiguous()
    ids = torch.full((a.steps + 4, P, K), -1, dtype=torch.int32, device=dev)
    claim = torch.full((P, K), -3, dtype=torch.int32, device=dev)
    second = [po[k][P:].contiguous() for k in ("nodes", "xy", "desc", "n")]
    torch.cuda.synchronize(dev)

    def assoc(i):
        ctx.associate(po["points4d"], n_map, po["c2"], w, h, second[0], second[1], second[2], second[3], offs, od, ids[i], claim=claim)

    def timed(fn, n):
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        ctx.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    if a.pmc_calibrate:
        ctx.debug_valu_calib()
        ctx.synchronize()
    pose(); assoc(a.steps + 3)
    res = {"pairs": P, "workload": "C3", "ms_frontend_pairs_pose": timed(lambda i: pose(), a.steps),
           "ms_associate_map_points": timed(assoc, a.steps), "map_points_per_pair": float(n_map.float().mean()),
           "claimed_per_pair": float((claim >= 0).float().sum(1).mean()), "mean_reprojection_inliers": float(po["n_inliers"].float().mean())}
    if not a.no_prof:
        ctx.prof_enable(True)
        ctx.prof_reset()
        pose(); assoc(a.steps + 2)
        rep = ctx.prof_report()
        ctx.prof_enable(False)
        res["scopes_ms"] = {k: round(v[0] / v[1], 4) for k, v in rep.items() if v[1] > 0}
    print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
