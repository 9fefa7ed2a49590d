"""Result-record file format (include/vslam/Ingest.h, vslam_amd/records.py): layout pinned byte by byte, and a
write -> read round trip."""
import struct

import numpy as np
import pytest

from vslam_amd import records


def sample():
    rng = np.random.default_rng(5)
    header = dict(width=1280, height=720, max_corners=2000, hypotheses=4096, threshold=10.0, seed=0x5EED0001)
    recs = []
    for i in range(4):
        n = [0, 1, 37, 500][i]
        recs.append(dict(first_frame=(1 << 33) + i, winner=[-1, 0, 17, 4095][i], inliers=n, score=np.float32(rng.normal()),
                         F=rng.normal(size=9).astype(np.float32), matches=rng.integers(0, 2000, (n, 2)).astype(np.int32)))
    return header, recs


def test_layout_is_the_documented_one(tmp_path):
    header, recs = sample()
    p = tmp_path / "r.bin"
    records.write_records(p, header, recs[:2])
    data = p.read_bytes()
    assert data[:8] == b"VSLAMREC"
    assert struct.unpack_from("<5I", data, 8) == (1, 1280, 720, 2000, 4096)
    assert struct.unpack_from("<f", data, 28)[0] == 10.0
    assert struct.unpack_from("<II", data, 32) == (0x5EED0001, 0)
    off = 40                                    # record 0: no matches
    assert struct.unpack_from("<Qii", data, off) == ((1 << 33), -1, 0)
    assert np.array_equal(np.frombuffer(data, "<f4", 9, off + 20), recs[0]["F"])
    assert struct.unpack_from("<I", data, off + 56)[0] == 0
    off += 60                                   # record 1: one match
    assert struct.unpack_from("<Qii", data, off) == ((1 << 33) + 1, 0, 1)
    assert struct.unpack_from("<I", data, off + 56)[0] == 1
    assert np.array_equal(np.frombuffer(data, "<i4", 2, off + 60), recs[1]["matches"][0])
    assert len(data) == 40 + 60 + 60 + 8


def test_round_trip_is_bit_preserving(tmp_path):
    header, recs = sample()
    recs[2]["F"][3] = np.float32(np.nan)
    recs[3]["score"] = np.float32(np.inf)
    p = tmp_path / "r.bin"
    records.write_records(p, header, recs)
    h2, r2 = records.read_records(p)
    assert h2 == header
    assert len(r2) == len(recs)
    for a, b in zip(recs, r2):
        assert (a["first_frame"], a["winner"], a["inliers"]) == (b["first_frame"], b["winner"], b["inliers"])
        assert np.float32(a["score"]).tobytes() == np.float32(b["score"]).tobytes()
        assert a["F"].tobytes() == b["F"].tobytes()
        assert np.array_equal(a["matches"], b["matches"])


def test_truncation_is_reported(tmp_path):
    header, recs = sample()
    p = tmp_path / "r.bin"
    records.write_records(p, header, recs)
    data = p.read_bytes()
    (tmp_path / "cut.bin").write_bytes(data[:-5])
    with pytest.raises(ValueError):
        records.read_records(tmp_path / "cut.bin")
    (tmp_path / "bad.bin").write_bytes(b"NOTAVSLM" + data[8:])
    with pytest.raises(ValueError):
        records.read_records(tmp_path / "bad.bin")
