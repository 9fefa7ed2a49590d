#!/usr/bin/env python3
"""Inputs for / outputs of tools/opencv_dump.cpp (the OpenCV-side vector dumper).

    python tools/opencv_case.py export <case.bin>              the seeded inputs of tests/golden/frontend_v1.npz / _v2_grid.npz
    python tools/opencv_case.py import <dump.bin> <out.npz>    the dumper's output as an npz (commit it as
                                                               tests/golden/opencv_v1.npz; tests/test_opencv_pin.py reads it)

Container: records { u32 name_len, name, u32 dtype (0 u8, 1 i32, 2 f32, 3 f64), u32 ndim, u32 dims[ndim], data }."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64}
CODE = {np.dtype(v): k for k, v in DT.items()}


def write_records(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<II", CODE[a.dtype], a.ndim))
            f.write(struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(a.tobytes())


def read_records(path):
    out, data, off = {}, open(path, "rb").read(), 0
    while off < len(data):
        (nl,) = struct.unpack_from("<I", data, off); off += 4
        name = data[off:off + nl].decode(); off += nl
        dt, nd = struct.unpack_from("<II", data, off); off += 8
        dims = struct.unpack_from("<%dI" % nd, data, off); off += 4 * nd
        n = int(np.prod(dims)) if nd else 1
        a = np.frombuffer(data, dtype=DT[dt], count=n, offset=off).reshape(dims).copy()
        off += a.nbytes
        out[name] = a
    return out


def case_inputs():
    G = np.load(os.path.join(ROOT, "tests", "golden", "frontend_v1.npz"))
    G2 = np.load(os.path.join(ROOT, "tests", "golden", "frontend_v2_grid.npz"))
    h, w = G["e_bgr"].shape[1:3]
    # the pose helpers' inputs: the golden pair's fundamental matrix and inlier matches, intrinsics as src/vslam.cpp:29-33
    K = np.array([[525.0, 0, w / 2.0], [0, 525.0, h / 2.0], [0, 0, 1]], np.float32)
    pm = G["p_matches"]
    # the photographic pairs of tests/golden/real_v1.npz (round 5), remade from the stored crops
    sys.path.insert(0, ROOT)
    from vslam_amd import synth
    R = np.load(os.path.join(ROOT, "tests", "golden", "real_v1.npz"))
    real = [synth.real_pair(R[f"crop{i}"], tuple(R["motions"][i])) for i in range(len(R["names"]))]
    return {
        "p_bgr": np.stack([a for a, _ in real] + [b for _, b in real]), "p_maxc": np.array([int(R["params"][0])], np.int32),
        "g_bgr": G2["g_bgr"], "g_grid": np.array([2, 2], np.int32),                    # make_golden.py: a 2 x 2 grid
        "t_F": G["p_F"].reshape(3, 3), "t_K": K,
        "t_p1": np.ascontiguousarray(G["e_xy0"][pm[:, 0]]), "t_p2": np.ascontiguousarray(G["e_xy1"][pm[:, 1]]),
        "e_bgr": G["e_bgr"], "e_maxc": np.array([150], np.int32),                     # make_golden.py: maxc = 150
        "m_d1": G["m_d1"], "m_d2": G["m_d2"],
        "r_p1": G["r_p1"], "r_p2": G["r_p2"], "r_pairs": G["r_pairs"], "r_fsets": G["r_fsets"],
        "r_thr": np.array([10.0], np.float32),
    }


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "export":
        write_records(sys.argv[2], case_inputs())
    elif len(sys.argv) == 4 and sys.argv[1] == "import":
        d = read_records(sys.argv[2])
        np.savez_compressed(sys.argv[3], **d)
        print(sys.argv[3], sorted(d))
    else:
        sys.exit(__doc__)
