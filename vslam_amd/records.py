"""Reader / writer for the per-pair result records the ingest layer writes (include/vslam/Ingest.h).

File layout (little endian): 40-byte header "VSLAMREC", u32 version, width, height, max_corners, hypotheses,
f32 threshold, u32 seed, u32 reserved; then per pair: u64 first_frame, i32 winner, i32 inliers, f32 score,
f32 F[9], u32 n, n x (i32, i32).
"""
import struct

import numpy as np

MAGIC = b"VSLAMREC"
_HEAD = struct.Struct("<8s5IfII")
_REC = struct.Struct("<Qiif9fI")


def write_records(path, header, records):
    """header: dict(width, height, max_corners, hypotheses, threshold, seed); records: iterable of dicts with
    first_frame, winner, inliers, score, F (9 floats), matches ((n, 2) int32)."""
    with open(path, "wb") as f:
        f.write(_HEAD.pack(MAGIC, 1, header["width"], header["height"], header["max_corners"], header["hypotheses"],
                           header["threshold"], header["seed"], 0))
        for r in records:
            m = np.ascontiguousarray(r["matches"], dtype="<i4").reshape(-1, 2)
            F = np.asarray(r["F"], dtype="<f4").reshape(9)
            f.write(_REC.pack(r["first_frame"], r["winner"], r["inliers"], np.float32(r["score"]), *F.tolist(), len(m)))
            f.write(m.tobytes())


def read_records(path):
    """-> (header dict, list of record dicts); F as float32[9] (bit-preserving), matches as int32[n, 2]."""
    with open(path, "rb") as f:
        data = f.read()
    magic, version, w, h, maxc, hyp, thr, seed, _ = _HEAD.unpack_from(data, 0)
    if magic != MAGIC or version != 1:
        raise ValueError("not a vslam record file (version 1)")
    header = dict(width=w, height=h, max_corners=maxc, hypotheses=hyp, threshold=thr, seed=seed)
    off, recs = _HEAD.size, []
    while off < len(data):
        if off + _REC.size > len(data):
            raise ValueError("truncated record")
        first, winner, inliers, score = struct.unpack_from("<Qiif", data, off)
        F = np.frombuffer(data, dtype="<f4", count=9, offset=off + 20).copy()
        (n,) = struct.unpack_from("<I", data, off + 56)
        off += _REC.size
        if n > maxc:
            raise ValueError("corrupt record: more matches than max_corners")
        if off + 8 * n > len(data):
            raise ValueError("truncated record")
        m = np.frombuffer(data, dtype="<i4", count=2 * n, offset=off).reshape(n, 2).copy()
        off += 8 * n
        recs.append(dict(first_frame=first, winner=winner, inliers=inliers, score=np.float32(score), F=F, matches=m))
    return header, recs
