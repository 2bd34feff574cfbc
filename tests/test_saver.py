"""CPU: ReplaySaver mirror (slam.jl_amd/saver.py <-> src/io/saver.jl): the set_frame_wc! rules, a save / load round trip, the error
exits of load!, the BSON container, and -- when a maintainer has run tests/golden/make_golden_julia.jl -- the files BSON.jl wrote."""
import os
import struct

import numpy as np
import pytest


def _wc(t):
    T = np.eye(4); T[:3, 3] = t
    return T


def _fill(s):
    s.set_frame_wc(7, _wc((1.0, 2.0, 3.0))); s.set_frame_wc(9, _wc((-4.0, 0.5, 6.0)))
    s.set_frame_wc(7, _wc((1.5, 2.5, 3.5))); s.set_frame_wc(12, _wc((0.0, 0.0, 10.0)))


def test_set_frame_wc_rules(slam_host):
    s = slam_host.ReplaySaver()
    _fill(s)
    assert s.ids == {7: 1, 9: 2, 12: 3}                                   # 1-based position ids, first-seen order (saver.jl:47-53)
    P = np.asarray(s.positions)
    assert P.dtype == np.float32 and P.shape == (3, 3)
    assert np.array_equal(P, np.array([[1.5, 3.5, 2.5], [-4.0, 6.0, 0.5], [0.0, 10.0, 0.0]], np.float32))     # (x, z, y), frame 7 overwritten
    R = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]]); T = np.eye(4); T[:3, :3] = R; T[:3, 3] = (3.0, 4.0, 5.0)
    s.set_frame_wc(1, T)                                                   # only the translation column matters: wc * [0 0 0 1]
    assert np.array_equal(s.positions[-1], np.array([3.0, 5.0, 4.0], np.float32))


def test_save_load_round_trip_and_errors(slam_host, tmp_path):
    s = slam_host.ReplaySaver(); _fill(s)
    d = str(tmp_path / "replay")
    s.save(d)
    assert sorted(os.listdir(d)) == ["ids.bson", "positions.bson"]
    t = slam_host.ReplaySaver().load(d)
    assert t.ids == s.ids and np.array_equal(np.asarray(t.positions), np.asarray(s.positions))
    # the container: a BSON document is <int32 total size> ... <0x00>, and the float payload is the raw little-endian array
    raw = open(os.path.join(d, "positions.bson"), "rb").read()
    assert struct.unpack("<i", raw[:4])[0] == len(raw) and raw[-1] == 0
    assert np.asarray(s.positions, "<f4").tobytes() in raw
    # ids.bson is BSON.jl's struct lowering of a Dict{Int64,Int64}: {tag: "struct", type: Dict{Int64,Int64}, data: [keys, values]},
    # each a tagged Int64 array with a binary payload
    from slam_jl_amd import saver as sv
    doc = sv._dec_doc(open(os.path.join(d, "ids.bson"), "rb").read())["ids"]
    assert doc["tag"] == "struct" and doc["type"]["name"] == ["Base", "Dict"] and [q["name"] for q in doc["type"]["params"]] == [["Core", "Int64"]] * 2
    assert [q["tag"] for q in doc["data"]] == ["array", "array"] and doc["data"][0]["size"] == [3]
    assert np.frombuffer(doc["data"][0]["data"], "<i8").tolist() == [7, 9, 12] and np.frombuffer(doc["data"][1]["data"], "<i8").tolist() == [1, 2, 3]
    # files of the pre-round-3 layout still load
    legacy = str(tmp_path / "legacy"); os.makedirs(legacy)
    open(os.path.join(legacy, "positions.bson"), "wb").write(open(os.path.join(d, "positions.bson"), "rb").read())
    open(os.path.join(legacy, "ids.bson"), "wb").write(sv._enc_doc({"ids": {"tag": "dict", "keys": [sv._I64(7), sv._I64(9), sv._I64(12)], "vals": [sv._I64(1), sv._I64(2), sv._I64(3)]}}))
    assert slam_host.ReplaySaver().load(legacy).ids == s.ids
    with pytest.raises(FileNotFoundError):
        slam_host.ReplaySaver().load(str(tmp_path / "nowhere"))
    os.remove(os.path.join(d, "ids.bson"))
    with pytest.raises(FileNotFoundError):
        slam_host.ReplaySaver().load(d)
    # an empty saver round-trips too
    e = str(tmp_path / "empty"); slam_host.ReplaySaver().save(e)
    z = slam_host.ReplaySaver().load(e)
    assert z.ids == {} and len(z.positions) == 0


def test_files_written_by_julia_when_present(slam_host):
    d = os.path.join(os.path.dirname(__file__), "golden", "julia_replay")
    if not os.path.isdir(d):
        pytest.skip("tests/golden/julia_replay absent: run tests/golden/make_golden_julia.jl with Julia + SLAM.jl to pin the BSON lowering")
    t = slam_host.ReplaySaver().load(d)
    s = slam_host.ReplaySaver(); _fill(s)
    assert t.ids == s.ids and np.array_equal(np.asarray(t.positions), np.asarray(s.positions))
    # byte-level: what save() writes against what BSON.jl wrote (Dict iteration order is Julia's: compare the decoded documents' structure,
    # and the bytes when the key order happens to agree)
    import tempfile
    from slam_jl_amd import saver as sv
    with tempfile.TemporaryDirectory() as tmp:
        s.save(tmp)
        for name in ("positions.bson", "ids.bson"):
            mine = sv._dec_doc(open(os.path.join(tmp, name), "rb").read()); theirs = sv._dec_doc(open(os.path.join(d, name), "rb").read())
            key = name.split(".")[0]
            assert mine[key]["tag"] == theirs[key]["tag"] and mine[key]["type"] == theirs[key]["type"], name
        if sv._dec_doc(open(os.path.join(d, "ids.bson"), "rb").read())["ids"]["data"][0].get("data") == sv._dec_doc(open(os.path.join(tmp, "ids.bson"), "rb").read())["ids"]["data"][0]["data"]:
            assert open(os.path.join(tmp, "ids.bson"), "rb").read() == open(os.path.join(d, "ids.bson"), "rb").read()
        assert open(os.path.join(tmp, "positions.bson"), "rb").read() == open(os.path.join(d, "positions.bson"), "rb").read()


def test_container_level_against_an_independent_bson_implementation(slam_host, tmp_path):
    """The BSON CONTAINER saver.py writes (document sizes, element type bytes, cstring keys, int32 / int64 / binary / array encodings) is
    decoded by the `bson` package of the image (pymongo's codec, written independently of this repository) and gives back the same
    values our own decoder reads; a document encoded by that package loads through our decoder.  What stays unpinned is BSON.jl's
    LOWERING of the two Julia values (the "tag" / "type" / "size" conventions), not the byte format."""
    bson = pytest.importorskip("bson")
    from slam_jl_amd import saver as sv
    s = slam_host.ReplaySaver(); _fill(s)
    d = str(tmp_path / "replay"); s.save(d)
    for name in ("positions.bson", "ids.bson"):
        raw = open(os.path.join(d, name), "rb").read()
        theirs = bson.decode(raw) if hasattr(bson, "decode") else bson.BSON(raw).decode()
        ours = sv._dec_doc(raw)

        def same(a, b):
            if isinstance(a, dict):
                return isinstance(b, dict) and list(a) == list(b) and all(same(a[k], b[k]) for k in a)
            if isinstance(a, (list, tuple)):
                return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
            if isinstance(a, (bytes, bytearray)) or isinstance(b, (bytes, bytearray)):
                return bytes(a) == bytes(b)
            return int(a) == int(b) if isinstance(a, (int, np.integer)) or hasattr(a, "__int__") and not isinstance(a, float) else a == b
        assert same(ours, theirs), name
    # and the other way round: a document their encoder writes, read by ours
    enc = bson.encode if hasattr(bson, "encode") else bson.BSON.encode
    doc = {"k": {"tag": "array", "size": [bson.Int64(2)], "data": bson.Binary(np.arange(2, dtype="<i8").tobytes())}, "n": 5}
    back = sv._dec_doc(bytes(enc(doc)))
    assert back["n"] == 5 and back["k"]["tag"] == "array" and int(back["k"]["size"][0]) == 2 and bytes(back["k"]["data"]) == np.arange(2, dtype="<i8").tobytes()
