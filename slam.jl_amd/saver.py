"""ReplaySaver (src/io/saver.jl:28-101): the pose sink of the reference -- camera positions per frame id, dumped as two BSON files
(`positions.bson`, `ids.bson`) for the visualiser's replay.

Host code, no device work: `set_frame_wc` is `set_frame_wc!` (saver.jl:41-55: position = (wc * [0, 0, 0, 1])[[1, 3, 2]] as Float32,
appended for a new frame id, overwritten for a known one; ids map frame id -> 1-based position index), `save` / `load` are
saver.jl:62-100.

File format.  BSON.jl (a Project.toml dependency, not vendored under /root/reference) writes `@save file name` as one BSON document
{name: lower(value)}.  PARITY UNPINNED: there is no BSON.jl here to check against; the container level (document / array / binary /
int32 / int64 / double / string elements, bsonspec.org 1.1) is exact, the lowering of the two Julia values is restated from the
package's published scheme (BSON.jl src/extensions.jl):
  positions :: Vector{Point3f0}   {tag: "array", type: {tag: "datatype", name: ["GeometryBasics", "Point"], params: [3, <Float32>]},
                                   size: [n], data: <binary, n x 3 little-endian float32>}      (an array of an isbits element type)
  ids :: Dict{Int64,Int64}        BSON.jl has no dictionary tag for non-String / Symbol keys: such a Dict goes through its generic struct
                                   path, `lower(x) = {tag: "struct", type: <typeof(x)>, data: structdata(x)}` with
                                   `structdata(d::Dict) = Any[collect(keys(d)), collect(values(d))]`, and each Vector{Int64} is again a
                                   tagged array: {tag: "struct", type: {tag: "datatype", name: ["Base", "Dict"], params: [<Int64>, <Int64>]},
                                   data: [{tag: "array", type: <Int64>, size: [n], data: <binary, n little-endian int64>}, {... values}]}
                                   -- what the reference's `@load ids_file ids` (saver.jl:92) rebuilds a Dict{Int64,Int64} from.
`load` reads that layout (binary or plain-array keys / values), the positions as the tagged array above or as a plain BSON array of
3-vectors, and the `{tag: "dict", keys, vals}` files this module wrote before round 3.  tests/golden/make_golden_julia.jl saves a
ReplaySaver from Julia into tests/golden/julia_replay/ so that a maintainer can pin (or correct) this byte for byte against the real package
(tests/test_saver.py compares the bytes when the directory exists).
"""
import os
import struct

import numpy as np

__all__ = ["ReplaySaver"]


# ---- minimal BSON (bsonspec.org): what the two files need ------------------------------------------------------------------------
def _cstr(s):
    return s.encode("utf-8") + b"\x00"


def _enc_value(v):
    if isinstance(v, bool):
        return b"\x08", b"\x01" if v else b"\x00"
    if isinstance(v, (int, np.integer)):
        v = int(v)
        return (b"\x10", struct.pack("<i", v)) if -2**31 <= v < 2**31 and not isinstance(v, _I64) else (b"\x12", struct.pack("<q", v))
    if isinstance(v, float):
        return b"\x01", struct.pack("<d", v)
    if isinstance(v, str):
        b = v.encode("utf-8") + b"\x00"
        return b"\x02", struct.pack("<i", len(b)) + b
    if isinstance(v, (bytes, bytearray)):
        return b"\x05", struct.pack("<i", len(v)) + b"\x00" + bytes(v)
    if isinstance(v, dict):
        return b"\x03", _enc_doc(v)
    if isinstance(v, (list, tuple)):
        return b"\x04", _enc_doc({str(i): x for i, x in enumerate(v)})
    if v is None:
        return b"\x0a", b""
    raise TypeError(f"BSON: cannot encode {type(v)}")


class _I64(int):
    """an integer that is always written as BSON int64 (Julia's Int64)"""


def _enc_doc(d):
    body = b"".join(t + _cstr(k) + payload for k, (t, payload) in ((k, _enc_value(v)) for k, v in d.items()))
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


def _dec_doc(b, off=0, as_list=False):
    size = struct.unpack_from("<i", b, off)[0]
    end = off + size - 1
    p = off + 4
    out = {}
    while p < end:
        t = b[p]; p += 1
        q = b.index(b"\x00", p)
        key = b[p:q].decode("utf-8"); p = q + 1
        if t == 0x01:
            val = struct.unpack_from("<d", b, p)[0]; p += 8
        elif t == 0x02:
            n = struct.unpack_from("<i", b, p)[0]; val = b[p + 4:p + 4 + n - 1].decode("utf-8"); p += 4 + n
        elif t in (0x03, 0x04):
            n = struct.unpack_from("<i", b, p)[0]; val = _dec_doc(b, p, as_list=(t == 0x04)); p += n
        elif t == 0x05:
            n = struct.unpack_from("<i", b, p)[0]; val = bytes(b[p + 5:p + 5 + n]); p += 5 + n
        elif t == 0x08:
            val = b[p] != 0; p += 1
        elif t == 0x0A:
            val = None
        elif t == 0x10:
            val = struct.unpack_from("<i", b, p)[0]; p += 4
        elif t == 0x12:
            val = struct.unpack_from("<q", b, p)[0]; p += 8
        else:
            raise ValueError(f"BSON: element type 0x{t:02x} not supported")
        out[key] = val
    if b[end] != 0:
        raise ValueError("BSON: document not terminated")
    return [out[str(i)] for i in range(len(out))] if as_list else out


def _datatype(path, params=()):
    return {"tag": "datatype", "name": list(path), "params": list(params)}


# ---- the saver ----------------------------------------------------------------------------------------------------------------------
class ReplaySaver:
    """saver.jl:28-35: ids (frame id -> 1-based position index) and positions (n x 3 float32: x, z, y of the camera centre)."""

    def __init__(self):
        self.ids = {}
        self.positions = []

    def set_frame_wc(self, frame_id, wc):
        """set_frame_wc! (saver.jl:41-55).  wc: 4 x 4 camera -> world."""
        wc = np.asarray(wc, dtype=np.float64).reshape(4, 4)
        pos = (wc @ np.array([0.0, 0.0, 0.0, 1.0]))[[0, 2, 1]].astype(np.float32)
        pid = self.ids.get(int(frame_id), -1)
        if pid == -1:
            self.positions.append(pos)
            self.ids[int(frame_id)] = len(self.positions)
        else:
            self.positions[pid - 1] = pos

    def save(self, save_dir):
        """save (saver.jl:62-73)."""
        os.makedirs(save_dir, exist_ok=True)
        P = np.asarray(self.positions, dtype="<f4").reshape(-1, 3)
        f32 = _datatype(["Core", "Float32"])
        pos_doc = {"positions": {"tag": "array", "type": _datatype(["GeometryBasics", "Point"], [3, f32]),
                                 "size": [_I64(len(P))], "data": P.tobytes()}}
        # insertion order = first-seen order of the frame ids.  Julia's Dict is serialised in HASH-SLOT order (BSON.jl lowers the struct's
        # keys / vals arrays as they sit in the table), so a Julia-written ids.bson holds the same (key, value) pairs in another order:
        # parity of ids.bson is SEMANTIC (load! rebuilds the Dict either way), not byte-level; positions.bson is byte-level.
        keys = list(self.ids)
        i64 = _datatype(["Core", "Int64"])
        vec = lambda a: {"tag": "array", "type": i64, "size": [_I64(len(a))], "data": np.asarray(a, dtype="<i8").tobytes()}
        ids_doc = {"ids": {"tag": "struct", "type": _datatype(["Base", "Dict"], [i64, i64]),
                           "data": [vec(keys), vec([self.ids[k] for k in keys])]}}
        with open(os.path.join(save_dir, "positions.bson"), "wb") as f:
            f.write(_enc_doc(pos_doc))
        with open(os.path.join(save_dir, "ids.bson"), "wb") as f:
            f.write(_enc_doc(ids_doc))

    def load(self, save_dir):
        """load! (saver.jl:80-100), same error conditions."""
        if not os.path.isdir(save_dir):
            raise FileNotFoundError(f"Directory `{save_dir}` does not exist.")
        pf, idf = os.path.join(save_dir, "positions.bson"), os.path.join(save_dir, "ids.bson")
        if not os.path.isfile(pf):
            raise FileNotFoundError(f"Positions file `{pf}` not found.")
        if not os.path.isfile(idf):
            raise FileNotFoundError(f"Ids file `{idf}` not found.")
        pos = _dec_doc(open(pf, "rb").read())["positions"]
        if isinstance(pos, dict) and pos.get("tag") == "array":
            P = np.frombuffer(pos["data"], dtype="<f4").reshape(-1, 3)
        else:                                                   # a plain BSON array of 3-element arrays
            P = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
        ids = _dec_doc(open(idf, "rb").read())["ids"]
        def ints(a):                                             # a tagged Vector{Int64} (binary payload) or a plain BSON array
            if isinstance(a, dict) and a.get("tag") == "array":
                return np.frombuffer(a["data"], dtype="<i8").tolist() if isinstance(a["data"], (bytes, bytearray)) else list(a["data"])
            return list(a)
        if "keys" in ids and "vals" in ids:                      # this module's own layout before round 3
            k, v = ids["keys"], ids["vals"]
        elif "data" in ids and isinstance(ids["data"], list) and len(ids["data"]) == 2:      # BSON.jl: structdata(d) = [keys, values]
            k, v = ints(ids["data"][0]), ints(ids["data"][1])
        else:
            raise ValueError("ids.bson: unknown dictionary layout")
        self.ids = {int(a): int(b) for a, b in zip(k, v)}
        self.positions = [p.copy() for p in P]
        return self
