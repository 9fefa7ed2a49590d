#!/usr/bin/env python3
"""End-to-end rate of the capture loop (vslam::run_sequence): raw BGR24 file -> page-locked buffers -> device ->
front-end on consecutive frames -> record file.  Unlike bench.py's `value` this includes the file reads and the
host-to-device copies.

--slots N runs vslam::run_sequence_devices with N contexts on device 0 (one reader + one copy stream + one compute
stream each), the one-GPU rehearsal of the several-device capture loop.

usage: ingest_bench.py [--frames 257] [--batch 64] [--slots 0] [--width 1280 --height 720 --keypoints 2000 --hyp 4096]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=257)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--keypoints", type=int, default=2000)
    ap.add_argument("--hyp", type=int, default=4096)
    ap.add_argument("--slots", type=int, default=0, help="0: run_sequence; N > 0: run_sequence_devices, N contexts on device 0")
    args = ap.parse_args()
    import numpy as np
    from vslam_amd import records, synth

    tmp = tempfile.mkdtemp(prefix="vslam_ingest_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    vid, rec = os.path.join(tmp, "clip.bgr"), os.path.join(tmp, "clip.rec")
    # a clip whose neighbouring frames match: 'last' / 'current' halves of synthetic pairs interleaved, repeated
    base = synth.frames_numpy(0x5EED0001, 8, args.width, args.height)
    clip = np.empty_like(base)
    clip[0::2], clip[1::2] = base[:8], base[8:]
    with open(vid, "wb") as f:
        for i in range(args.frames):
            f.write(clip[i % 16].tobytes())
    out = {}
    for label in ("warm-up", "timed"):
        t0 = time.perf_counter()
        try:
            frames, pairs, secs, _ = records.run_sequence(vid, rec, args.width, args.height, args.batch, args.keypoints, args.hyp,
                                                       10.0, 1, devices=[0] * args.slots if args.slots > 0 else None)
        except RuntimeError as e:
            sys.exit(str(e))
        out = {"frames": frames, "pairs": pairs, "loop_seconds": secs,
               "wall_seconds": time.perf_counter() - t0, "pairs_per_s": pairs / secs,
               "input_GB_per_s": frames * args.width * args.height * 3 / secs / 1e9,
               "batch_frames": args.batch, "slots": args.slots, "record_bytes": os.path.getsize(rec)}
    os.remove(vid)
    os.remove(rec)
    os.rmdir(tmp)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
