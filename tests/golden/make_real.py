#!/usr/bin/env python3
"""Generates tests/golden/real_v1.npz: PHOTOGRAPHIC frame pairs and the oracle's outputs for them.

Every other input of the test-suite is synthetic texture or noise.  Photographs have what those lack: saturated plateaus
(night sky, blown highlights), smooth gradients (skin, porcelain), JPEG block structure, near-ties between neighbouring
responses -- where a from-memory detail of an OpenCV routine, or a certified margin of the two-tier corner detector, is
most likely to be wrong in a way synthetic data never shows.

Sources: four public-domain / CC0 photographs that ship with scikit-image as FILES in the build container (read with PIL;
nothing of scikit-image is imported, and only the pixels travel):
    astronaut.png          NASA, "no known copyright restrictions, released into the public domain"
    coffee.png             CC0 by the photographer (Rachel Michetti)
    rocket.jpg             SpaceX, released in the public domain
    hubble_deep_field.jpg  NASA / HubbleSite, public domain
Each is brought to at least 760 x 600 by an integer-arithmetic bilinear resize (vslam_amd.synth.resample_fixed: the same
bytes everywhere), cropped to the central 736 x 576, and stored.  The pair's frames are windows of that crop
(synth.real_pair): frame A the central 640 x 480, frame B the same window after a small camera motion (rotation <= 1.5
degrees, shift <= 12 px), resampled at sub-pixel positions.  The file holds the crops, the motions, a CRC of every frame
(so that a test knows it has remade the same bytes) and what the oracle computes: per frame the keypoints, descriptors and
k-d tree of extract_features (src/Frame.cpp:53-80, max_corners 1000), per pair match_features' output (src/Frame.cpp:82-105;
RansacFilter(8, 512, 10) seeded 0xC0DE ^ pair).

    python tests/golden/make_real.py [directory with the four files]
"""
import os
import sys
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vslam_amd import synth  # noqa: E402

SOURCES = [("astronaut.png", (1.2, 7.0, -4.0)), ("coffee.png", (-0.8, -11.0, 5.0)), ("rocket.jpg", (0.5, 3.0, 9.0)),
           ("hubble_deep_field.jpg", (-1.5, 12.0, -6.0))]
MAXC, HYP, THR, SEED = 1000, 512, 10.0, 0xC0DE


def crop_of(path):
    rgb = np.asarray(Image.open(path).convert("RGB"))
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    H, W = bgr.shape[:2]
    s = max(760.0 / W, 600.0 / H, 1.0)
    if s > 1.0:
        nw, nh = int(np.ceil(W * s)), int(np.ceil(H * s))
        bgr = synth.resample_fixed(bgr, nh, nw, 1.0 / s, 0.0, 0.0, 1.0 / s, 0.0, 0.0)
        H, W = nh, nw
    y0, x0 = (H - 576) // 2, (W - 736) // 2
    return np.ascontiguousarray(bgr[y0:y0 + 576, x0:x0 + 736])


def main():
    src_dir = sys.argv[1] if len(sys.argv) > 1 else "/opt/conda/lib/python3.9/site-packages/skimage/data"
    o = Oracle()
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation()
    g = {"names": np.array([n for n, _ in SOURCES]), "motions": np.array([m for _, m in SOURCES], np.float64),
         "params": np.array([MAXC, HYP, SEED], np.int64), "threshold": np.array([THR], np.float32)}
    crcs = []
    for i, (name, motion) in enumerate(SOURCES):
        crop = crop_of(os.path.join(src_dir, name))
        g[f"crop{i}"] = crop
        a, b = synth.real_pair(crop, motion)
        crcs += [zlib.crc32(a.tobytes()), zlib.crc32(b.tobytes())]
        fa = o.extract_features(a, MAXC, ca, sa, pat)
        fb = o.extract_features(b, MAXC, ca, sa, pat)
        r = o.match_features(fa["xy"], fa["desc"], fb["xy"], fb["desc"], SEED ^ i, HYP, THR)
        for tag, f in (("a", fa), ("b", fb)):
            g[f"xy_{tag}{i}"], g[f"desc_{tag}{i}"], g[f"nodes_{tag}{i}"] = f["xy"], f["desc"], f["nodes"]
            g[f"ndet_{tag}{i}"] = np.array([f["n_detected"]], np.int32)
        g[f"matches{i}"] = r["matches"]
        g[f"F{i}"] = np.asarray(r["F"], np.float32)
        g[f"prelim{i}"] = np.array([r["prelim"], r["rc"]], np.int32)
        print(f"{name}: {fa['n']} / {fb['n']} keypoints, {r['prelim']} preliminary, {len(r['matches'])} inlier matches (rc {r['rc']})")
    g["frame_crc32"] = np.array(crcs, np.uint32)
    out = os.path.join(HERE, "real_v1.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
