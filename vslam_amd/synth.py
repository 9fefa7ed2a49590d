"""Seeded synthetic inputs: descriptor sets, point correspondences, frames.

Everything is generated from an integer seed with numpy's PCG64 (identical on every machine) so
the CPU oracle and the GPU path see the same bytes.  Frames for the throughput bench can also be
generated directly on the device with torch (same construction, device RNG).
"""
import math

import numpy as np


def descriptors_pair(seed, k1, k2, match_frac=0.6, flip_p=0.05):
    """Frame-A descriptors = uniform random bytes; frame B = a permutation of A with each bit
    flipped w.p. flip_p for match_frac of the rows and fresh random rows for the rest."""
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 256, size=(k1, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, size=(k2, 32), dtype=np.uint8)
    n_match = int(min(k1, k2) * match_frac)
    src = rng.permutation(k1)[:n_match]
    dst = rng.permutation(k2)[:n_match]
    flips = rng.random((n_match, 256)) < flip_p
    d2[dst] = d1[src] ^ np.packbits(flips, axis=1)
    truth = np.full(k1, -1, dtype=np.int64)
    truth[src] = dst
    return d1, d2, truth


def two_view_points(seed, k, width, height, inlier_frac=0.7, noise_px=0.5, integer=True):
    """k correspondences between two pinhole views (f = 525, principal point at the centre, as
    src/vslam.cpp:29-33) of random 3-D points under a small rigid motion; the rest are uniform
    outliers.  Coordinates are integer-valued like Shi-Tomasi output."""
    rng = np.random.default_rng(seed)
    f, cx, cy = 525.0, width / 2.0, height / 2.0
    ang = np.deg2rad(rng.uniform(-1.5, 1.5, size=3))
    Rx = np.array([[1, 0, 0], [0, math.cos(ang[0]), -math.sin(ang[0])], [0, math.sin(ang[0]), math.cos(ang[0])]])
    Ry = np.array([[math.cos(ang[1]), 0, math.sin(ang[1])], [0, 1, 0], [-math.sin(ang[1]), 0, math.cos(ang[1])]])
    Rz = np.array([[math.cos(ang[2]), -math.sin(ang[2]), 0], [math.sin(ang[2]), math.cos(ang[2]), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = rng.uniform(-0.3, 0.3, size=3)
    p1 = np.zeros((k, 2)); p2 = np.zeros((k, 2))
    p1[:, 0] = rng.uniform(0, width - 1, size=k); p1[:, 1] = rng.uniform(0, height - 1, size=k)
    depth = rng.uniform(2.0, 12.0, size=k)
    X = np.stack([(p1[:, 0] - cx) / f * depth, (p1[:, 1] - cy) / f * depth, depth], axis=1)
    X2 = X @ R.T + t
    p2[:, 0] = f * X2[:, 0] / X2[:, 2] + cx + rng.normal(0, noise_px, size=k)
    p2[:, 1] = f * X2[:, 1] / X2[:, 2] + cy + rng.normal(0, noise_px, size=k)
    outl = rng.random(k) > inlier_frac
    p2[outl, 0] = rng.uniform(0, width - 1, size=outl.sum()); p2[outl, 1] = rng.uniform(0, height - 1, size=outl.sum())
    p2[:, 0] = np.clip(p2[:, 0], 0, width - 1); p2[:, 1] = np.clip(p2[:, 1], 0, height - 1)
    if integer:
        p1 = np.rint(p1); p2 = np.rint(p2)
    return p1.astype(np.float32), p2.astype(np.float32), ~outl


def brief_pattern():
    """ORB's learned rBRIEF table (OpenCV `bit_pattern_31_`, what cv::ORB::compute samples: src/Frame.cpp:57,68):
    256 x (x0, y0, x1, y1) int8.  Shipped as data (`brief_pattern_31.npy`, written by tools/make_brief_pattern.py);
    the same bytes are compiled into the library (`vslam_brief_pattern_31()`, the meaning of a NULL d_pattern)."""
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "brief_pattern_31.npy")).copy()


def synthetic_pattern(seed=0xB21EF):
    """A seeded Gaussian 256 x (x0,y0,x1,y1) int8 pattern (sigma = patch/5, clipped to +-13): NOT ORB's table.  For
    tests that want descriptors under other tables than the default one."""
    rng = np.random.default_rng(seed)
    p = np.clip(np.rint(rng.normal(0, 31 / 5.0, size=(256, 4))), -13, 13).astype(np.int8)
    same = (p[:, 0] == p[:, 2]) & (p[:, 1] == p[:, 3])
    p[same, 2] = np.where(p[same, 2] < 13, p[same, 2] + 1, p[same, 2] - 1)
    return p


def keypoint_rotation(angle_deg=-1.0):
    """(cos, sin) of the KeyPoint angle as float32: cv::KeyPoint(p, 20) leaves angle = -1 and
    ORB::compute does not recompute it for provided keypoints."""
    a = np.float32(angle_deg) * np.float32(math.pi / 180.0)
    return float(np.float32(math.cos(float(a)))), float(np.float32(math.sin(float(a))))


def frames_numpy(seed, n_pairs, width, height):
    """BGR uint8 frames, shape (2*n_pairs, H, W, 3): frames [0,n) are 'last', [n,2n) 'current'.
    Blocky random texture (three cell sizes) so corner detectors find thousands of junctions,
    low-amplitude noise, and frame B = frame A translated by a few pixels with fresh noise, except for
    an independently moving block (about a fifth of the image) whose matches are outliers to the
    dominant epipolar geometry."""
    rng = np.random.default_rng(seed)
    out = np.zeros((2 * n_pairs, height, width, 3), dtype=np.uint8)
    for p in range(n_pairs):
        base = np.zeros((height + 32, width + 32), dtype=np.float32)
        for cell, amp in ((16, 70.0), (24, 50.0), (10, 25.0)):
            gh, gw = (height + 32) // cell + 2, (width + 32) // cell + 2
            g = rng.uniform(-amp, amp, size=(gh, gw)).astype(np.float32)
            up = np.kron(g, np.ones((cell, cell), dtype=np.float32))
            oy, ox = rng.integers(0, cell, size=2)
            base += up[oy:oy + height + 32, ox:ox + width + 32]
        base = np.clip(128 + base, 8, 247)
        dx, dy = rng.integers(-12, 13, size=2)
        a = base[16:16 + height, 16:16 + width]
        b = base[16 + dy:16 + dy + height, 16 + dx:16 + dx + width].copy()
        dx2, dy2 = rng.integers(-12, 13, size=2)
        bw, bh = int(width * 0.45), int(height * 0.45)
        bx, by = rng.integers(0, width - bw), rng.integers(0, height - bh)
        b[by:by + bh, bx:bx + bw] = base[16 + dy2 + by:16 + dy2 + by + bh, 16 + dx2 + bx:16 + dx2 + bx + bw]
        for f, img in ((p, a), (n_pairs + p, b)):
            for c in range(3):
                noise = rng.integers(-6, 7, size=(height, width))
                tint = (c - 1) * 4
                out[f, :, :, c] = np.clip(img + noise + tint, 0, 255).astype(np.uint8)
    return out


def frames_torch(seed, n_pairs, width, height, device):
    """Same construction as frames_numpy with torch's RNG on `device` (bench inputs only)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    H2, W2 = height + 32, width + 32
    base = torch.zeros((n_pairs, H2, W2), dtype=torch.float32, device=device)
    for cell, amp in ((16, 70.0), (24, 50.0), (10, 25.0)):
        gh, gw = H2 // cell + 2, W2 // cell + 2
        grid = (torch.rand((n_pairs, gh, gw), generator=g, device=device) * 2 - 1) * amp
        up = grid.repeat_interleave(cell, dim=1).repeat_interleave(cell, dim=2)
        base += up[:, 3:3 + H2, 5:5 + W2]
    base = torch.clamp(128 + base, 8, 247)
    out = torch.empty((2 * n_pairs, height, width, 3), dtype=torch.uint8, device=device)
    shifts = torch.randint(-12, 13, (n_pairs, 4), generator=g, device=device).cpu()
    bw, bh = int(width * 0.45), int(height * 0.45)
    corner = torch.stack([torch.randint(0, width - bw, (n_pairs,), generator=g, device=device),
                          torch.randint(0, height - bh, (n_pairs,), generator=g, device=device)], 1).cpu()
    for p in range(n_pairs):
        dx, dy, dx2, dy2 = (int(v) for v in shifts[p])
        bx, by = int(corner[p, 0]), int(corner[p, 1])
        a = base[p, 16:16 + height, 16:16 + width]
        b = base[p, 16 + dy:16 + dy + height, 16 + dx:16 + dx + width].clone()
        b[by:by + bh, bx:bx + bw] = base[p, 16 + dy2 + by:16 + dy2 + by + bh, 16 + dx2 + bx:16 + dx2 + bx + bw]
        for f, img in ((p, a), (n_pairs + p, b)):
            noise = torch.randint(-6, 7, (height, width, 3), generator=g, device=device).float()
            tint = torch.tensor([-4.0, 0.0, 4.0], device=device)
            out[f] = torch.clamp(img[:, :, None] + noise + tint, 0, 255).to(torch.uint8)
    return out


def frames_torch_hard(seed, n_pairs, width, height, device, outlier_area=0.55):
    """The harder regime SURVEY.md 8(d) specifies, on `device`: frame B is frame A seen after a small camera motion —
    an in-plane rotation of up to 1.5 degrees about the image centre and a shift of up to 12 px (the infinite
    homography) plus a parallax of 0..10 px along one epipolar direction that depends on which of four depth layers a
    96-px cell belongs to — resampled bilinearly at sub-pixel positions, with fresh noise; so one fundamental matrix
    explains every true correspondence and the scene is not a single plane.  On `outlier_area` of the image (cells
    drawn at random) B instead shows the texture displaced by an unrelated vector of up to 40 px (matches there are
    outliers to the dominant geometry: 40-45 % of the matches) or, for a quarter of those cells, texture that A does
    not contain at all.
    Returns uint8 (2 * n_pairs, H, W, 3): frames [0, n) are 'last', [n, 2n) 'current', like frames_torch."""
    import torch
    import torch.nn.functional as Fn
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) ^ 0x4A7D)
    pad = 64
    H2, W2 = height + 2 * pad, width + 2 * pad
    out = torch.empty((2 * n_pairs, height, width, 3), dtype=torch.uint8, device=device)
    ys, xs = torch.meshgrid(torch.arange(height, device=device, dtype=torch.float32),
                            torch.arange(width, device=device, dtype=torch.float32), indexing="ij")
    cell = 96
    ch, cw = (height + cell - 1) // cell, (width + cell - 1) // cell
    tint = torch.tensor([-4.0, 0.0, 4.0], device=device)
    chunk = 16   # pairs per texture batch (memory)
    for p0 in range(0, n_pairs, chunk):
        nb = min(chunk, n_pairs - p0)
        base = torch.zeros((nb + 1, H2, W2), dtype=torch.float32, device=device)   # the last one: texture A never sees
        for c, amp in ((16, 70.0), (24, 50.0), (10, 25.0)):
            gh, gw = H2 // c + 2, W2 // c + 2
            grid = (torch.rand((nb + 1, gh, gw), generator=g, device=device) * 2 - 1) * amp
            up = grid.repeat_interleave(c, dim=1).repeat_interleave(c, dim=2)
            base += up[:, 3:3 + H2, 5:5 + W2]
        base = torch.clamp(128 + base, 8, 247)
        par = torch.rand((nb, 8), generator=g, device=device).cpu()
        for i in range(nb):
            p = p0 + i
            theta = float(par[i, 0] * 2 - 1) * 1.5 * math.pi / 180.0
            sx, sy = float(par[i, 1] * 2 - 1) * 12.0, float(par[i, 2] * 2 - 1) * 12.0
            phi = float(par[i, 3]) * 2 * math.pi
            ex, ey = math.cos(phi), math.sin(phi)
            # per-cell motion class in B's pixel domain
            layer = torch.randint(0, 4, (ch, cw), generator=g, device=device)
            depth_shift = torch.tensor([0.0, 3.0, 6.5, 10.0], device=device)[layer]
            r = torch.rand((ch, cw), generator=g, device=device)
            is_out = r < outlier_area
            is_new = r < outlier_area / 4.0
            ov = (torch.rand((ch, cw, 2), generator=g, device=device) * 2 - 1) * 40.0
            dx_c = torch.where(is_out, ov[..., 0], depth_shift * ex)
            dy_c = torch.where(is_out, ov[..., 1], depth_shift * ey)
            up = lambda t: t.repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:height, :width]
            dxm, dym, newm = up(dx_c), up(dy_c), up(is_new)
            # where B's pixel (x, y) comes from in A's frame: undo the rotation and shift, then the layer's displacement
            cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
            ct, st = math.cos(theta), math.sin(theta)
            ux, uy = xs - cx - sx, ys - cy - sy
            srcx = ct * ux + st * uy + cx - dxm
            srcy = -st * ux + ct * uy + cy - dym
            gx = (srcx + pad) / (W2 - 1) * 2 - 1
            gy = (srcy + pad) / (H2 - 1) * 2 - 1
            grid = torch.stack([gx, gy], -1)[None]
            b_img = Fn.grid_sample(base[i][None, None], grid, mode="bilinear", padding_mode="border", align_corners=True)[0, 0]
            n_img = Fn.grid_sample(base[nb][None, None], grid, mode="bilinear", padding_mode="border", align_corners=True)[0, 0]
            b_img = torch.where(newm, n_img, b_img)
            a_img = base[i, pad:pad + height, pad:pad + width]
            for f, img in ((p, a_img), (n_pairs + p, b_img)):
                noise = torch.randint(-6, 7, (height, width, 3), generator=g, device=device).float()
                out[f] = torch.clamp(img[:, :, None] + noise + tint, 0, 255).to(torch.uint8)
    return out


def resample_fixed(img, height, width, a, b, c, d, tx, ty):
    """Bilinear resampling of an (H, W, 3) uint8 image in INTEGER arithmetic (16.16 fixed point): output pixel (x, y) reads
    the source at (a x + b y + tx, c x + d y + ty), clamped to the image.  The same bytes on every machine -- this is how
    tests/golden/real_v1.npz's second frames are remade from its first frames (tests/golden/make_real.py) -- and the
    sub-pixel interpolation gives the smooth, slightly blurred content photographs have after any resampling."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    Q = 1 << 16
    fx = (np.int64(round(a * Q)) * np.arange(width, dtype=np.int64)[None, :] + np.int64(round(b * Q)) * np.arange(height, dtype=np.int64)[:, None]
          + np.int64(round(tx * Q)))
    fy = (np.int64(round(c * Q)) * np.arange(width, dtype=np.int64)[None, :] + np.int64(round(d * Q)) * np.arange(height, dtype=np.int64)[:, None]
          + np.int64(round(ty * Q)))
    fx = np.clip(fx, 0, (W - 1) * Q)
    fy = np.clip(fy, 0, (H - 1) * Q)
    x0, y0 = fx >> 16, fy >> 16
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    wx, wy = (fx & (Q - 1)) >> 8, (fy & (Q - 1)) >> 8          # 8-bit weights
    src = img.astype(np.int64)
    out = np.empty((height, width, img.shape[2]), np.uint8)
    for ch in range(img.shape[2]):
        p = src[:, :, ch]
        top = p[y0, x0] * (256 - wx) + p[y0, x1] * wx
        bot = p[y1, x0] * (256 - wx) + p[y1, x1] * wx
        out[:, :, ch] = ((top * (256 - wy) + bot * wy + (1 << 15)) >> 16).astype(np.uint8)
    return out


def real_pair(first, motion):
    """The two 640 x 480 BGR frames of a photographic pair: both are windows of the source photograph `first` (H, W, 3; at
    least 720 x 560), the second after a small camera motion (rotation in degrees about the window centre, shift in pixels)."""
    rot, dx, dy = motion
    H, W = first.shape[:2]
    x0, y0 = (W - 640) // 2, (H - 480) // 2
    a = resample_fixed(first, 480, 640, 1.0, 0.0, 0.0, 1.0, x0, y0)
    th = np.deg2rad(rot)
    ca, sa = float(np.cos(th)), float(np.sin(th))
    cx, cy = x0 + 320.0, y0 + 240.0
    # source position of output (x, y): rotate (x - 320, y - 240) by rot, add centre + shift
    b = resample_fixed(first, 480, 640, ca, -sa, sa, ca, cx + dx - 320 * ca + 240 * sa, cy + dy - 320 * sa - 240 * ca)
    return a, b


def frames_torch_photo(seed, n_pairs, width, height, device, npz=None):
    """A batch with PHOTOGRAPHIC statistics (bench.py --data photo): every frame is a window of one of the four public-domain
    photographs of tests/golden/real_v1.npz (736 x 576 crops: saturated plateaus, smooth gradients, JPEG block structure),
    magnified by a per-pair factor of 1.0 .. 1.4 and continued by mirror images of itself where the window leaves the crop
    (so any frame size can be cut; content stays photographic, nothing is synthesised).  Frame B of a pair is frame A after a
    small camera motion -- rotation up to 1.5 degrees about the image centre, shift up to 12 px -- resampled at sub-pixel
    positions.  Bilinear resampling in integer arithmetic (16.16 fixed point, 8-bit weights, as resample_fixed): the same bytes
    on every device.  Returns uint8 (2 * n_pairs, H, W, 3): frames [0, n) are 'last', [n, 2n) 'current'."""
    import os
    import torch
    if npz is None:
        npz = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_v1.npz")
    g = np.load(npz)
    crops = [torch.from_numpy(g[f"crop{i}"]).to(device) for i in range(4)]
    out = torch.empty((2 * n_pairs, height, width, 3), dtype=torch.uint8, device=device)
    Q = 1 << 16
    xs = torch.arange(width, device=device, dtype=torch.int64)[None, :]
    ys = torch.arange(height, device=device, dtype=torch.int64)[:, None]
    rng = np.random.default_rng(int(seed) ^ 0xF070)

    def sample(img, a, b, c, d, tx, ty):
        H, W = img.shape[:2]
        fx = int(round(a * Q)) * xs + int(round(b * Q)) * ys + int(round(tx * Q))
        fy = int(round(c * Q)) * xs + int(round(d * Q)) * ys + int(round(ty * Q))
        # mirror continuation: reflect the 16.16 coordinate into [0, (n - 1) Q] (period 2 (n - 1) Q)
        px, py = 2 * (W - 1) * Q, 2 * (H - 1) * Q
        fx = torch.remainder(fx, px)
        fx = torch.where(fx > (W - 1) * Q, px - fx, fx)
        fy = torch.remainder(fy, py)
        fy = torch.where(fy > (H - 1) * Q, py - fy, fy)
        x0, y0 = fx >> 16, fy >> 16
        x1, y1 = torch.clamp(x0 + 1, max=W - 1), torch.clamp(y0 + 1, max=H - 1)
        wx, wy = ((fx & (Q - 1)) >> 8)[..., None], ((fy & (Q - 1)) >> 8)[..., None]
        src = img.to(torch.int64)
        top = src[y0, x0] * (256 - wx) + src[y0, x1] * wx
        bot = src[y1, x0] * (256 - wx) + src[y1, x1] * wx
        return ((top * (256 - wy) + bot * wy + (1 << 15)) >> 16).to(torch.uint8)

    for p in range(n_pairs):
        img = crops[p % 4]
        zoom = 1.0 + 0.4 * rng.random()
        ox, oy = rng.uniform(-200, 400), rng.uniform(-150, 300)     # where the window sits (may leave the crop: mirrored)
        rot = np.deg2rad(rng.uniform(-1.5, 1.5))
        dx, dy = rng.uniform(-12, 12), rng.uniform(-12, 12)
        k = 1.0 / zoom
        out[p] = sample(img, k, 0.0, 0.0, k, ox, oy)
        ca, sa = float(np.cos(rot)), float(np.sin(rot))
        cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
        # B's pixel (x, y) shows A's position R (x - c) + c + shift; A's position (u, v) shows the crop at (k u + ox, k v + oy)
        a_, b_, c_, d_ = k * ca, -k * sa, k * sa, k * ca
        tx = k * (cx + dx - ca * cx + sa * cy) + ox
        ty = k * (cy + dy - sa * cx - ca * cy) + oy
        out[n_pairs + p] = sample(img, a_, b_, c_, d_, tx, ty)
    return out
