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


def run_sequence(video_path, record_path, width, height, batch_frames, max_corners, hypotheses, threshold, seed,
                 max_frames=0, devices=None):
    """vslam::run_sequence (devices None) / vslam::run_sequence_devices (a list of device indices, one context each) through
    their C entry points in libvslam_host.so: raw BGR24 file in, record file out.  -> (frames, pairs, seconds,
    batches redone with whole-image corner lists after a VSLAM_ERR_CAPACITY).
    Raises RuntimeError with the library's message."""
    import ctypes
    from . import build
    lib = ctypes.CDLL(build.build_host())
    frames, pairs, secs = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
    err = ctypes.create_string_buffer(512)
    head = (str(video_path).encode(), str(record_path).encode(), int(width), int(height), int(batch_frames), int(max_corners),
            int(hypotheses), ctypes.c_float(threshold), ctypes.c_uint32(seed & 0xFFFFFFFF), ctypes.c_uint64(max_frames))
    tail = (ctypes.byref(frames), ctypes.byref(pairs), ctypes.byref(secs), err, 512)
    if devices is None:
        rc = lib.vslam_host_run_sequence(*head, *tail)
    else:
        devices = list(devices)
        dev = (ctypes.c_int * max(len(devices), 1))(*devices)
        rc = lib.vslam_host_run_sequence_devices(*head, dev, len(devices), *tail)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    lib.vslam_host_last_batches_redone.restype = ctypes.c_uint64
    return frames.value, pairs.value, secs.value, int(lib.vslam_host_last_batches_redone())
