"""Per-kernel times of one extraction stage alone on the C3 batch (HIP events on the kernels' stream):
python tools/stage_time.py [good_features|extract] [reps].  For kernel A/B work; prints name, ms per launch."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vslam_amd import Context, synth  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "good_features"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    w, h, K, P = 1280, 720, 2000, int(sys.argv[3]) if len(sys.argv) > 3 else 256
    ctx = Context(0)
    bgr = synth.frames_torch(0x5EED0002, P, w, h, "cuda")
    gray = ctx.bgr2gray(bgr)
    pat = torch.from_numpy(synth.brief_pattern()).cuda()
    ca, sa = synth.keypoint_rotation()

    def run():
        if what == "good_features":
            return ctx.good_features(gray, K)
        return ctx.extract_features(bgr, K, ca, sa, pat)
    out = run()
    ctx.synchronize()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(reps):
        out = run()
    ctx.synchronize()
    for name, (ms, cnt) in ctx.prof_report().items():
        print(f"{name:28s} {ms / max(cnt, 1):8.4f} ms  x{cnt}")
    n = out[1] if what == "good_features" else out["n"]
    print("corners per frame: mean", float(n.float().mean()))


if __name__ == "__main__":
    main()
