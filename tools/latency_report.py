#!/usr/bin/env python3
"""Single-frame latency through the reference's own API (the drop-in C++ surfaces) next to the oracle on the same
host, at the reference's parameters: 1280x720, 3000 corners, rf(8, 100, 10).  Writes one JSON object.

usage: python tools/latency_report.py [out.json]"""
import json
import os
import subprocess
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vslam_amd import build, synth  # noqa: E402


def main():
    w, h = 1280, 720
    bgr = synth.frames_numpy(0x1A7E, 1, w, h)
    pat = synth.brief_pattern()
    tmp = tempfile.mkdtemp()
    fin = os.path.join(tmp, "in.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("2i", w, h))
        f.write(bgr.tobytes())
        f.write(pat.tobytes())
    build.build_host()
    exe = os.path.join(tmp, "latency_demo")
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tools", "latency_demo.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "vslam_amd"), "-lvslam_host", "-lvslam_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "vslam_amd")], check=True)
    dev = json.loads(subprocess.run([exe, fin, "20"], check=True, capture_output=True, text=True, timeout=600).stdout)

    from oracle_lib import Oracle
    o = Oracle()
    ca, sa = synth.keypoint_rotation()
    feats, t_ext = [], []
    for rep in range(3):
        for i in range(2):
            t0 = time.perf_counter()
            r = o.extract_features(bgr[i], 3000, ca, sa, pat)
            t_ext.append((time.perf_counter() - t0) * 1e6)
            if rep == 0:
                feats.append(r)
    t_match = []
    for rep in range(5):
        t0 = time.perf_counter()
        m = o.match_features(feats[0]["xy"], feats[0]["desc"], feats[1]["xy"], feats[1]["desc"], 1, 100, 10.0)
        t_match.append((time.perf_counter() - t0) * 1e6)
    out = {
        "what": "one call at a time through include/vslam/{Frame,KDTree,RansacFilter}.h (device) vs the oracle (CPU port, one thread), "
                "same host, 1280x720 synthetic frame, 3000 corners, RansacFilter(8, 100, 10); medians, microseconds",
        "device_adapters": dev,
        "oracle_cpu": {"extract_features_us": float(np.median(t_ext)), "match_features_us": float(np.median(t_match)),
                       "keypoints": [int(feats[0]["n"]), int(feats[1]["n"])], "inlier_matches": int(len(m["matches"])),
                       "radius_search_us_per_query": "0.13 (reference's own src/KDTree.cpp:145-171 measured in SURVEY.md section 6; "
                                                     "the oracle's C entry point relinks the tree per call and is not a timing proxy)"},
    }
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
