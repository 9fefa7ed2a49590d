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


def test_cpp_reader_and_writer_agree_with_python(tmp_path):
    """The C++ RecordWriter / RecordReader (libvslam_host.so) and vslam_amd.records read each other's files."""
    import os
    import subprocess
    from vslam_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build.build_host()
    exe = str(tmp_path / "records_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "native", "records_check.cpp"),
                    "-I" + os.path.join(root, "include"), "-L" + os.path.join(root, "vslam_amd"), "-lvslam_host", "-lvslam_amd",
                    "-Wl,-rpath," + os.path.join(root, "vslam_amd")], check=True)
    # C++ writes, Python reads
    f1 = str(tmp_path / "cpp.bin")
    subprocess.run([exe, "write", f1], check=True, timeout=60)
    head, recs = records.read_records(f1)
    assert head == dict(width=640, height=480, max_corners=500, hypotheses=512, threshold=10.0, seed=0xC0FFEE)
    assert [r["first_frame"] for r in recs] == [1000, 1001, 1002]
    for i, r in enumerate(recs):
        assert (r["winner"], r["inliers"], float(r["score"])) == (i - 1, 10 * i, 0.5 * i)
        assert np.array_equal(r["F"], (np.arange(9) + 9 * i).astype(np.float32) * 0.125)
        assert np.array_equal(r["matches"], np.array([[k, 2 * k + i] for k in range(10 * i)], np.int32).reshape(-1, 2))
    # Python writes, C++ reads
    header, mine = sample()
    f2 = str(tmp_path / "py.bin")
    records.write_records(f2, header, mine)
    out = subprocess.run([exe, "read", f2], check=True, timeout=60, capture_output=True, text=True).stdout.split("\n")
    assert out[0].split()[:6] == ["H", "1", "1280", "720", "2000", "4096"] and int(out[0].split()[7]) == 0x5EED0001
    assert float.fromhex(out[0].split()[6]) == 10.0
    for line, r in zip(out[1:], mine):
        t = line.split()
        assert t[0] == "R" and int(t[1]) == r["first_frame"] and int(t[2]) == r["winner"] and int(t[3]) == r["inliers"]
        assert int(t[4], 16) == int(np.float32(r["score"]).view(np.uint32)) and int(t[5]) == len(r["matches"])
        assert [int(x, 16) for x in t[6:15]] == [int(v) for v in r["F"].view(np.uint32)]
        acc = 0
        for a, b in r["matches"]:
            acc = acc * 31 + int(a) * 7 + int(b)
            acc = (acc + 2 ** 63) % 2 ** 64 - 2 ** 63       # the C++ side accumulates in a wrapping int64
        assert int(t[15]) == acc
    assert len([l for l in out if l.startswith("R")]) == len(mine)
    # a truncated file makes the C++ reader fail loudly
    cut = str(tmp_path / "cut.bin")
    open(cut, "wb").write(open(f2, "rb").read()[:-3])
    assert subprocess.run([exe, "read", cut], capture_output=True).returncode == 1
    # so does a corrupt match count (it must not be trusted as an allocation size)
    blob = bytearray(open(f2, "rb").read())
    blob[40 + 56:40 + 60] = (0x7FFFFFF0).to_bytes(4, "little")       # first record's n
    bad = str(tmp_path / "bad_n.bin")
    open(bad, "wb").write(bytes(blob))
    r = subprocess.run([exe, "read", bad], capture_output=True, text=True)
    assert r.returncode == 1 and "corrupt" in r.stderr
    with pytest.raises(ValueError):
        records.read_records(bad)
