"""Prototype of the two-tier corner response (not product code): the integer-exact cheap value U~ and its certified
margin against the oracle's cornerMinEigenVal, on synthetic and adversarial images.  Prints how many pixels each tier
would touch.  Usage: python tools/me_proto.py [w h]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from vslam_amd import synth  # noqa: E402

U24 = 2.0 ** -24
SCALE = 1.0 / (4 * 3 * 255.0)
C0 = 0.5 * SCALE * SCALE


def refl(a, axis, lo, hi):
    """pad by REFLECT_101"""
    return np.pad(a, [(lo, hi) if i == axis else (0, 0) for i in range(a.ndim)], mode="reflect")


def cheap(gray):
    g = gray.astype(np.int64)
    gp = np.pad(g, 1, mode="reflect")
    hx = gp[:, 2:] - gp[:, :-2]                      # (h+2, w)
    rs = gp[:, :-2] + 2 * gp[:, 1:-1] + gp[:, 2:]    # (h+2, w)
    dx = hx[:-2] + 2 * hx[1:-1] + hx[2:]
    dy = rs[2:] - rs[:-2]
    cxx, cxy, cyy = dx * dx, dx * dy, dy * dy

    def box(c):
        p = np.pad(c, 1, mode="reflect")
        r = p[:, :-2] + p[:, 1:-1] + p[:, 2:]
        return r[:-2] + r[1:-1] + r[2:]
    A, B, C = box(cxx), box(cxy), box(cyy)
    tr = (A + C).astype(np.float32)
    d = (A - C).astype(np.float32)
    b2 = (2 * B).astype(np.float32)
    t = (d * d).astype(np.float32)
    t = (b2.astype(np.float64) * b2 + t).astype(np.float32)
    U = tr - np.sqrt(t).astype(np.float32)
    return U.astype(np.float32), tr


def margin(tr):
    return 0.016 * np.sqrt(tr.astype(np.float64)) + 32 * U24 * tr + 1e-3


def max8(a, fill=-np.inf):
    p = np.pad(a, 1, constant_values=fill)
    h, w = a.shape
    out = np.full(a.shape, -np.inf, a.dtype)
    for dy in range(3):
        for dx in range(3):
            if dy == 1 and dx == 1:
                continue
            out = np.maximum(out, p[dy:dy + h, dx:dx + w])
    return out


def analyse(name, gray, o):
    eig = o.min_eigen(gray)
    U, tr = cheap(gray)
    Uo = eig.astype(np.float64) / C0
    err = np.abs(Uo - U)
    m = margin(tr)
    ratio = (err / m).max()
    mx = eig.max()
    thr = np.float32(np.float64(mx) * 0.01)
    h, w = gray.shape
    inner = np.zeros_like(eig, bool)
    inner[1:-1, 1:-1] = True
    cand = inner & (eig > thr) & ~(max8(eig) > eig)
    hi, lo = U + m, U - m
    thrU = thr / C0 * (1 - 1e-6)
    poss = inner & (hi > thrU) & (hi >= max8(lo))
    cert = poss & (lo >= max8(hi))
    # uniform-margin variant
    M = margin(np.float32(tr.max()))
    possM = inner & (U + M > thrU) & (U + 2 * M >= max8(U))
    certM = possM & (U - 2 * M >= max8(U))
    print(f"{name}: max e {mx:.3e} thrU {thrU:.1f} trmax {tr.max():.3g} err/margin max {ratio:.3f} (err max {err.max():.3g}) "
          f"cand {cand.sum()} possible {poss.sum()} uncertain {poss.sum() - cert.sum()} | uniform M={M:.1f}: possible {possM.sum()} uncertain {possM.sum() - certM.sum()}")
    assert not (cand & ~poss).any()
    assert ratio < 1.0
    return ratio


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
    o = Oracle()
    bgr = synth.frames_numpy(0x5EED0002, 2, w, h)
    for f in range(2):
        analyse(f"synth{f}", o.bgr2gray(bgr[f]), o)
    rng = np.random.default_rng(1)
    analyse("uniform noise", rng.integers(0, 256, (h, w), dtype=np.uint8), o)
    analyse("binary noise", (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8), o)
    analyse("bright noise", rng.integers(250, 256, (h, w), dtype=np.uint8), o)
    chk = ((np.add.outer(np.arange(h) // 7, np.arange(w) // 5) & 1) * 255).astype(np.uint8)
    analyse("checker", chk, o)
    analyse("checker+1", np.clip(chk.astype(int) + rng.integers(-1, 2, (h, w)), 0, 255).astype(np.uint8), o)
    ramp = ((np.add.outer(np.arange(h) * 3, np.arange(w) * 2)) % 256).astype(np.uint8)
    analyse("ramp", ramp, o)
    lowc = (128 + rng.integers(-2, 3, (h, w))).astype(np.uint8)
    analyse("low contrast", lowc, o)


if __name__ == "__main__":
    main()
