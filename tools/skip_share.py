#!/usr/bin/env python3
"""how many 256-pixel row segments of the corner detector's strips could skip the square root /
maxima / queue block because a certified upper bound of the response lies below the threshold?  Best case for the skip: the
FINAL threshold (0.01 x the frame's maximum response; the kernel only knows a running lower bound of it) and the trace
tr = a + c as the bound (lambda_min = tr - sqrt(...) <= tr).  CPU only (numpy restatement of cornerMinEigenVal's sums).
    python tools/skip_share.py > profiles/r06_detector_skip_share.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vslam_amd import synth  # noqa: E402

o = Oracle()


def box(a):
    ap = np.pad(a, 1, mode="reflect")
    s = np.zeros_like(a)
    for i in range(3):
        for j in range(3):
            s += ap[i:i + a.shape[0], j:j + a.shape[1]]
    return s


def share(frames):
    per = []
    for f in frames:
        g = o.bgr2gray(f).astype(np.float64)
        gp = np.pad(g, 1, mode="reflect")
        dx = (gp[:-2, 2:] + 2 * gp[1:-1, 2:] + gp[2:, 2:]) - (gp[:-2, :-2] + 2 * gp[1:-1, :-2] + gp[2:, :-2])
        dy = (gp[2:, :-2] + 2 * gp[2:, 1:-1] + gp[2:, 2:]) - (gp[:-2, :-2] + 2 * gp[:-2, 1:-1] + gp[:-2, 2:])
        sc = 1.0 / (4 * 3 * 255.0)
        dx *= sc
        dy *= sc
        A, C, B = 0.5 * box(dx * dx), 0.5 * box(dy * dy), box(dx * dy)
        tr = A + C
        lam = tr - np.sqrt((A - C) ** 2 + B * B)
        thr = 0.01 * lam.max()
        h, w = tr.shape
        n = s = 0
        for x0 in range(0, w, 256):
            seg = tr[:, x0:x0 + 256].max(axis=1)
            n += h
            s += int((seg < thr).sum())
        per.append(s / n)
    return per


if __name__ == "__main__":
    dev = torch.device("cpu")
    ph = share(synth.frames_torch_photo(0x5EED0001, 8, 1280, 720, dev).numpy()[:8])
    hd = share(synth.frames_torch_hard(0x5EED0001, 2, 1280, 720, dev).numpy()[:2])
    ez = share(synth.frames_torch(0x5EED0001, 2, 1280, 720, dev).numpy()[:2])
    print("share of 256-pixel row segments whose trace bound lies below the FINAL threshold (best case for a wave-uniform skip)")
    print(f"  photographic regime (8 frames of bench.py --data photo): {np.mean(ph):.3f}   per frame {[round(x, 2) for x in ph]}")
    print(f"  hard regime (SURVEY 8(d) data):                          {np.mean(hd):.3f}")
    print(f"  easy regime:                                             {np.mean(ez):.3f}")
    print("criterion for building the skip (VERDICT r5 #8): >= 0.20 on the photographs -> not met, not built")
