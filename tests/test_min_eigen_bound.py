"""The certified bound behind the two-tier corner response (vslam_amd/csrc/response.hip, min_eigen_tiered_kernel).

The kernel decides from a cheap value U~ (integer Sobel responses, exact 3x3 sums, one float square root) which pixels
need the oracle's exact arithmetic; that is only sound if |e / c0 - U~| <= m(tr) for every pixel.  This test restates
U~ and m in numpy and measures the distance to the oracle's cornerMinEigenVal on synthetic frames and on adversarial
images (saturated noise, bright low-contrast noise, checkerboards, ramps, flat).  It also checks the two set
inclusions the kernel relies on: every true candidate and the frame maximum are among the 'possible' pixels.
"""
import numpy as np
import pytest

from vslam_amd import synth

U24 = 2.0 ** -24
C0 = 0.5 / (4 * 3 * 255.0) ** 2


def cheap(gray):
    """U~ and tr exactly as the kernel forms them (float32 where the kernel rounds)."""
    g = gray.astype(np.int64)
    gp = np.pad(g, 1, mode="reflect")                     # BORDER_REFLECT_101
    hx = gp[:, 2:] - gp[:, :-2]
    rs = gp[:, :-2] + 2 * gp[:, 1:-1] + gp[:, 2:]
    dx = hx[:-2] + 2 * hx[1:-1] + hx[2:]
    dy = rs[2:] - rs[:-2]

    def box(c):
        p = np.pad(c, 1, mode="reflect")
        r = p[:, :-2] + p[:, 1:-1] + p[:, 2:]
        return r[:-2] + r[1:-1] + r[2:]
    A, B, C = box(dx * dx), box(dx * dy), box(dy * dy)
    assert A.max() < 2 ** 24 and C.max() < 2 ** 24 and np.abs(B).max() < 2 ** 23
    tr = (A + C).astype(np.float32)
    d = (A - C).astype(np.float32)
    b2 = (2 * B).astype(np.float32)
    t = (b2.astype(np.float64) * b2 + (d * d).astype(np.float32)).astype(np.float32)   # one fma
    return (tr - np.sqrt(t).astype(np.float32)).astype(np.float32), tr


def margin(tr):
    return 0.016 * np.sqrt(tr.astype(np.float64)) + 32 * U24 * tr + 1e-3


def max8(a):
    p = np.pad(a, 1, constant_values=-np.inf)
    h, w = a.shape
    out = np.full(a.shape, -np.inf)
    for dy in range(3):
        for dx in range(3):
            if dy != 1 or dx != 1:
                out = np.maximum(out, p[dy:dy + h, dx:dx + w])
    return out


def images():
    w, h = 320, 200
    rng = np.random.default_rng(7)
    bgr = synth.frames_numpy(0x5EED0002, 1, w, h)
    yield "synthetic", None, bgr
    yield "uniform noise", rng.integers(0, 256, (h, w), dtype=np.uint8), None
    yield "saturated noise", (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8), None
    yield "bright low-contrast noise", rng.integers(250, 256, (h, w), dtype=np.uint8), None
    yield "mid low-contrast noise", (128 + rng.integers(-2, 3, (h, w))).astype(np.uint8), None
    chk = ((np.add.outer(np.arange(h) // 7, np.arange(w) // 5) & 1) * 255).astype(np.uint8)
    yield "checkerboard", chk, None
    yield "checkerboard + 1", np.clip(chk.astype(int) + rng.integers(-1, 2, (h, w)), 0, 255).astype(np.uint8), None
    yield "ramp", (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8), None
    yield "flat", np.full((h, w), 200, np.uint8), None
    yield "one bright pixel", np.where(np.add.outer(np.arange(h) == 50, np.arange(w) == 60) > 1, 255, 3).astype(np.uint8), None


def test_cheap_response_is_within_its_certified_margin(oracle):
    worst = 0.0
    for name, gray, bgr in images():
        if gray is None:
            gray = oracle.bgr2gray(bgr[0])
        eig = oracle.min_eigen(gray)
        U, tr = cheap(gray)
        err = np.abs(eig.astype(np.float64) / C0 - U)
        m = margin(tr)
        ratio = float((err / m).max())
        worst = max(worst, ratio)
        assert ratio < 0.25, (name, ratio)     # measured: below 0.04; the bound itself is 1.0

        # the inclusions the kernel needs, with the margin it uses (2 m of the largest tr nearby)
        mx = eig.max()
        thr = np.float32(np.float64(mx) * 0.01)
        inner = np.zeros(eig.shape, bool)
        inner[1:-1, 1:-1] = True
        cand = inner & (eig > thr) & ~(max8(eig) > eig)
        trp = np.pad(tr, 1, mode="edge")
        hh, ww = tr.shape
        tm = np.max([trp[i:i + hh, j:j + ww] for i in range(3) for j in range(3)], axis=0)
        m2 = 2 * margin(tm)
        thrU = thr / C0 * (1 - 2.0 ** -19)
        possible = (U + m2 >= np.maximum(max8(U), thrU))
        assert not (cand & ~possible).any(), name
        certain = possible & (U - m2 >= max8(U))
        assert not (certain & inner & (max8(eig) > eig)).any(), name      # a certain maximum is a maximum
        assert (U + m2)[np.unravel_index(np.argmax(eig), eig.shape)] >= (U - m2 / 2).max(), name   # the maximum queues
    assert worst > 0.0


def test_gray_from_two_byte_dot_products():
    """response.hip (gray_x2_16): cvtColor's (3735 b + 19235 g + 9798 r + 2^14) >> 15 formed as two 4 x u8 dot products
    with the DOUBLED weights split into bytes, 256 * (29, 150, 76) + (46, 70, 140), and 2 * 2^14 = 256 * 128 as the first
    one's addend: the gray value is byte 2 of the sum and byte 3 stays zero — for every (b, g, r)."""
    b, g, r = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    ref = (3735 * b + 19235 * g + 9798 * r + (1 << 14)) >> 15
    hi = 29 * b + 150 * g + 76 * r + 128
    s = 46 * b + 70 * g + 140 * r + (hi << 8)
    assert int(s.max()) < (1 << 24)
    assert np.array_equal((s >> 16) & 0xFF, ref)
    assert np.array_equal(s >> 24, np.zeros_like(s))
