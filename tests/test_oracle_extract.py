"""Independent sanity checks of the extract half of the oracle (oracle/vso_extract.cpp).

The reference runs these steps inside OpenCV (src/Frame.cpp:56 cvtColor, :61 goodFeaturesToTrack, :68 ORB::compute),
OpenCV is not installed, and the golden vectors come from the oracle itself -- so nothing else would catch a gross
from-memory error in the restatement.  Each check below recomputes a step from its textbook definition with
numpy / scipy in float64 (different code, different arithmetic) and holds the oracle to it within a stated tolerance.
These are sanity bounds, not bit parity: the bit-level pin against a real OpenCV stays tests/test_opencv_pin.py.
"""
import numpy as np
import pytest
from scipy import ndimage

from vslam_amd import synth


def frames(seed, w, h, n=1):
    return synth.frames_numpy(seed, n, w, h)


def images():
    rng = np.random.default_rng(11)
    yield "blocks", frames(21, 192, 160)[0]
    yield "blocks2", frames(22, 131, 97)[1]
    yield "noise", rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    step = np.zeros((80, 100, 3), np.uint8)
    step[:, 50:] = 255
    step[40:, :] = 255 - step[40:, :]
    yield "checker-corner", step
    # photographs (tests/golden/real_v1.npz: public-domain images, round 5): windows with a saturated night sky, skin and
    # porcelain gradients, JPEG block structure -- content the synthetic textures do not have
    import os
    real = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_v1.npz"))
    for i, (y0, x0) in enumerate([(40, 60), (200, 300), (100, 10), (300, 200)]):
        yield f"photo-{str(real['names'][i]).split('.')[0]}", np.ascontiguousarray(real[f"crop{i}"][y0:y0 + 150, x0:x0 + 200])


# ------------------------------------------------------------------------------------------------ cvtColor (:56)
@pytest.mark.parametrize("name,bgr", list(images()))
def test_bgr2gray_is_the_bt601_luma(oracle, name, bgr):
    g = oracle.bgr2gray(bgr).astype(np.float64)
    want = 0.114 * bgr[..., 0] + 0.587 * bgr[..., 1] + 0.299 * bgr[..., 2]
    assert np.abs(g - want).max() <= 0.5 + 1e-2, name         # a correctly rounded fixed-point luma (weights to 2^-15)
    assert np.abs(g - np.rint(want)).max() <= 1, name


def test_bgr2gray_channel_order_and_extremes(oracle):
    px = np.zeros((1, 4, 3), np.uint8)
    px[0, 0] = (255, 0, 0); px[0, 1] = (0, 255, 0); px[0, 2] = (0, 0, 255); px[0, 3] = (255, 255, 255)
    assert oracle.bgr2gray(px)[0].tolist() == [29, 150, 76, 255]      # B, G, R weights 0.114 / 0.587 / 0.299


# ------------------------------------------------------------------------------- cornerMinEigenVal (inside :61)
def min_eigen_f64(gray):
    """Textbook Shi-Tomasi response: 3x3 Sobel derivatives scaled by 1 / (4 * blockSize * 255), products summed over the
    3x3 block (not averaged), smaller eigenvalue of 1/2 [[Sxx, Sxy], [Sxy, Syy]] ... as OpenCV defines it:
    (a + c) - sqrt((a - c)^2 + b^2) with a = Sxx / 2, b = Sxy, c = Syy / 2.  REFLECT_101 = scipy's 'mirror'."""
    g = gray.astype(np.float64)
    scale = 1.0 / (4 * 3 * 255.0)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float64) * scale
    dx = ndimage.correlate(g, kx, mode="mirror")
    dy = ndimage.correlate(g, kx.T, mode="mirror")
    box = np.ones((3, 3))
    sxx = ndimage.correlate(dx * dx, box, mode="mirror")
    sxy = ndimage.correlate(dx * dy, box, mode="mirror")
    syy = ndimage.correlate(dy * dy, box, mode="mirror")
    a, b, c = 0.5 * sxx, sxy, 0.5 * syy
    return (a + c) - np.sqrt((a - c) ** 2 + b * b)


@pytest.mark.parametrize("name,bgr", list(images()))
def test_min_eigen_against_float64_definition(oracle, name, bgr):
    gray = oracle.bgr2gray(bgr)
    e = oracle.min_eigen(gray).astype(np.float64)
    want = min_eigen_f64(gray)
    scale = np.abs(want).max()
    assert scale > 0, name
    # float32 pipeline vs float64: 1e-5 of the frame's largest response (the subtraction cancels, so small responses
    # carry the absolute error of the large terms)
    err = np.abs(e - want)
    assert err.max() <= 1e-5 * scale, (name, err.max(), scale)
    # and where the response is a usable corner value the agreement is relative to the value itself
    strong = want > 0.01 * scale
    assert strong.sum() > 10, name
    assert (err[strong] / want[strong]).max() <= 2e-4, name


def test_min_eigen_known_values(oracle):
    flat = np.full((20, 20), 77, np.uint8)
    assert np.all(oracle.min_eigen(flat) == 0)                 # no gradient, no response
    edge = np.zeros((20, 20), np.uint8); edge[:, 10:] = 200
    assert np.abs(oracle.min_eigen(edge)).max() < 1e-9         # a straight edge has one zero eigenvalue
    corner = np.zeros((21, 21), np.uint8); corner[10:, 10:] = 200
    e = oracle.min_eigen(corner)
    y, x = np.unravel_index(np.argmax(e), e.shape)
    assert abs(y - 10) <= 1 and abs(x - 10) <= 1 and e.max() > 1e-4      # an L corner responds at the corner


# ---------------------------------------------------------------------------------- goodFeaturesToTrack (:61)
def greedy_reference(eig, max_corners, quality, min_dist):
    """goodFeaturesToTrack's selection from its definition, O(N^2): pixels above quality * max that equal the maximum of
    their 3x3 neighbourhood, image border excluded, strongest first (ties: larger address first), a candidate is taken
    unless an already taken corner lies closer than min_dist (Euclidean, strict)."""
    h, w = eig.shape
    thr = np.float32(np.float64(eig.max()) * quality)
    v = np.where(eig > thr, eig, np.float32(0))
    dil = ndimage.maximum_filter(v, size=3, mode="constant", cval=0.0)
    ys, xs = np.nonzero((v != 0) & (v == dil))
    inner = (ys >= 1) & (ys < h - 1) & (xs >= 1) & (xs < w - 1)
    ys, xs = ys[inner], xs[inner]
    addr = ys * w + xs
    order = np.lexsort((-addr, -v[ys, xs].astype(np.float64)))
    taken = []
    for i in order:
        x, y = int(xs[i]), int(ys[i])
        if min_dist >= 1 and any((x - tx) ** 2 + (y - ty) ** 2 < min_dist * min_dist for tx, ty in taken):
            continue
        taken.append((x, y))
        if len(taken) == max_corners:
            break
    return np.array(taken, np.float32).reshape(-1, 2)


@pytest.mark.parametrize("name,bgr", list(images()))
@pytest.mark.parametrize("maxc,min_dist", [(60, 3.0), (400, 3.0), (150, 7.0), (100, 1.0), (50, 0.0)])
def test_good_features_is_the_greedy_selection(oracle, name, bgr, maxc, min_dist):
    gray = oracle.bgr2gray(bgr)
    got = oracle.good_features(gray, maxc, min_dist=min_dist)
    want = greedy_reference(oracle.min_eigen(gray), maxc, 0.01, min_dist)
    assert got.shape == want.shape and np.array_equal(got, want), (name, maxc, min_dist, len(got), len(want))
    if len(got) > 1 and min_dist >= 1:
        d = got[:, None, :] - got[None, :, :]
        d2 = (d ** 2).sum(-1) + np.eye(len(got)) * 1e9
        assert d2.min() >= min_dist * min_dist
    assert np.array_equal(got, np.rint(got)), "corner coordinates are integer-valued floats"


def test_good_features_plateau_tie_order(oracle):
    """Equal responses: the larger address goes first (greaterThanPtr compares pointers when values tie)."""
    img = np.zeros((64, 64), np.uint8)
    for (y, x) in ((16, 16), (16, 40), (40, 16), (40, 40)):
        img[y:y + 8, x:x + 8] = 255                            # four identical squares -> identical corner responses
    got = oracle.good_features(img, 8, min_dist=3.0)
    want = greedy_reference(oracle.min_eigen(img), 8, 0.01, 3.0)
    assert np.array_equal(got, want)
    e = oracle.min_eigen(img)
    v = e[got[:, 1].astype(int), got[:, 0].astype(int)]
    assert np.all(np.diff(v) <= 0)
    same = np.nonzero(np.diff(v) == 0)[0]
    assert len(same) > 0
    addr = got[:, 1] * 64 + got[:, 0]
    assert np.all(addr[same] > addr[same + 1])


# ---------------------------------------------------------------------------------- GaussianBlur 7x7, sigma 2 (:68)
@pytest.mark.parametrize("name,bgr", list(images()))
def test_gaussian7_against_scipy(oracle, name, bgr):
    gray = oracle.bgr2gray(bgr)
    got = oracle.gaussian7(gray).astype(np.float64)
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / (2 * 2.0 ** 2)); k /= k.sum()          # getGaussianKernel(7, 2)
    f = ndimage.correlate1d(gray.astype(np.float64), k, axis=1, mode="mirror")
    f = ndimage.correlate1d(f, k, axis=0, mode="mirror")
    # Q8 taps deviate from the real kernel by up to 0.0032 each: worst case ~ 2 * 7 * 0.0032 * 255 / 2; measured far below
    assert np.abs(got - f).max() <= 1.5, (name, np.abs(got - f).max())
    assert np.abs(got - np.rint(f)).max() <= 1, name
    assert abs((got - f).mean()) < 0.15, name                 # no brightness bias: the taps sum to exactly 1


def test_gaussian7_taps(oracle):
    """An impulse of 255 in a black image comes back as the outer product of the taps: they sum to 256 (flat areas stay
    flat), are symmetric, decrease from the centre, and stay within 1/256 of the real Gaussian's."""
    img = np.zeros((31, 31), np.uint8); img[15, 15] = 255
    out = oracle.gaussian7(img).astype(np.float64)
    assert out[:12].sum() == 0 and out[19:].sum() == 0 and out[:, :12].sum() == 0 and out[:, 19:].sum() == 0
    flat = np.full((40, 40), 201, np.uint8)
    assert np.all(oracle.gaussian7(flat) == 201)
    row = np.zeros((1, 64), np.uint8); row[0, 32] = 255
    wide = np.repeat(row, 32, axis=0)                          # constant along y: the column pass is the identity
    taps = oracle.gaussian7(wide)[16, 29:36].astype(np.float64) / 255.0
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / 8.0); k /= k.sum()
    assert np.allclose(taps, taps[::-1]) and np.all(np.diff(taps[:4]) > 0)
    assert np.abs(taps - k).max() < 1.0 / 256 + 0.5 / 255, (taps, k)


# ------------------------------------------------------------------------------- ORB::compute: border, rBRIEF (:68)
def test_border_filter_is_31_pixels(oracle):
    w, h = 200, 150
    blur = np.random.default_rng(5).integers(0, 256, (h, w), dtype=np.uint8)
    xs, ys = np.meshgrid(np.arange(24, 40), np.arange(24, 40))
    pts = np.concatenate([np.stack([xs.ravel(), ys.ravel()], 1),
                          np.stack([w - 1 - xs.ravel(), h - 1 - ys.ravel()], 1)]).astype(np.float32)
    ca, sa = synth.keypoint_rotation()
    _, keep = oracle.orb_describe(blur, pts, ca, sa, synth.brief_pattern())
    inside = (pts[:, 0] >= 31) & (pts[:, 0] < w - 31) & (pts[:, 1] >= 31) & (pts[:, 1] < h - 31)   # Rect::contains
    assert np.array_equal(keep, np.nonzero(inside)[0])
    assert inside.sum() not in (0, len(pts))


@pytest.mark.parametrize("angle", [-1.0, 0.0, 37.0])
def test_rbrief_bits_from_the_definition(oracle, angle):
    """Steered BRIEF: bit i of the descriptor = I(c + R p_i0) < I(c + R p_i1) on the blurred image, offsets rotated by
    the keypoint angle and rounded to the nearest pixel, bit i in byte i // 8 at position i % 8."""
    w, h = 160, 120
    blur = oracle.gaussian7(np.random.default_rng(8).integers(0, 256, (h, w), dtype=np.uint8))
    rng = np.random.default_rng(9)
    pts = np.stack([rng.integers(31, w - 31, 50), rng.integers(31, h - 31, 50)], 1).astype(np.float32)
    pat = synth.brief_pattern()
    ca, sa = synth.keypoint_rotation(angle)
    desc, keep = oracle.orb_describe(blur, pts, ca, sa, pat)
    assert len(keep) == 50
    p = pat.astype(np.float64)
    hits = total = 0
    for k in range(50):
        cx, cy = int(pts[k, 0]), int(pts[k, 1])
        bits = np.unpackbits(desc[k], bitorder="little")
        for i in range(256):
            v = []
            frac = []
            for e in (0, 2):
                rx = p[i, e] * ca - p[i, e + 1] * sa
                ry = p[i, e] * sa + p[i, e + 1] * ca
                frac += [abs(rx - np.floor(rx) - 0.5), abs(ry - np.floor(ry) - 0.5)]
                v.append(int(blur[cy + int(np.rint(ry)), cx + int(np.rint(rx))]))
            if min(frac) < 1e-4:
                continue                                       # a rotated offset on a rounding tie: float32 decides
            total += 1
            hits += int(bits[i] == (v[0] < v[1]))
    assert total > 12000 and hits == total
