"""Host mirror of the reference's KITTI odometry reader (reference: example/kitty/kitty.jl:1-102) plus the 8-bit frame
path into the device: frames stay uint8 from the PNG to `slam_pyr_update_u8` /
`slam_pyr_update_batch_u8_dev` (the reference converts to Gray{Float64} on the host, example/kitty/main.jl:36-41; the
bytes/255 conversion happens on the GPU instead, bit-identical).

`KittyDataset(base_dir, sequence, stereo=True)` keeps the reference's fields (K, Ti0, poses, timestamps, frame
directories) and indexing (`dataset[i]` -> (left, right); 0-based here, 1-based in Julia).  `write_poses` emits the
KITTI pose text format (12 numbers per line), the dataset's own interchange format; the reference's ReplaySaver
(src/io/saver.jl) is mirrored in saver.py."""
import os
import struct
import zlib

import numpy as np


def parse_matrix(line):
    """12 numbers -> 4x4 with [0 0 0 1] appended (kitty.jl:1-4)."""
    m = np.array([float(t) for t in line.split()], dtype=np.float64)
    if m.size != 12:
        raise ValueError(f"expected 12 numbers, got {m.size}")
    return np.vstack([m.reshape(3, 4), [0.0, 0.0, 0.0, 1.0]])


def read_poses(poses_file):
    with open(poses_file) as f:
        return [parse_matrix(ln) for ln in f if ln.strip()]


def read_timestamps(timestamps_file):
    with open(timestamps_file) as f:
        return [float(ln) for ln in f if ln.strip()]


def write_poses(path, poses):
    """KITTI odometry result format: one line per frame, the top 3x4 of the 4x4 pose, row-major."""
    with open(path, "w") as f:
        for T in poses:
            f.write(" ".join(f"{v:.9e}" for v in np.asarray(T, dtype=np.float64)[:3, :4].reshape(-1)) + "\n")


# ---- 8-bit grayscale PNG (what KITTI odometry ships); PIL when present, a zlib reader otherwise ------------------
def _png_chunks(buf):
    if buf[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    p = 8
    while p < len(buf):
        n, tag = struct.unpack(">I4s", buf[p:p + 8])
        yield tag, buf[p + 8:p + 8 + n]
        p += 12 + n


def decode_png_gray8(buf):
    """bytes of a non-interlaced 8-bit grayscale PNG -> (H, W) uint8."""
    W = H = None
    idat = []
    for tag, data in _png_chunks(buf):
        if tag == b"IHDR":
            W, H, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", data)
            if depth != 8 or ctype != 0 or interlace != 0:
                raise ValueError(f"only 8-bit non-interlaced grayscale PNGs are supported (depth {depth}, colour type {ctype})")
        elif tag == b"IDAT":
            idat.append(data)
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(H, W + 1)
    out = np.zeros((H, W), dtype=np.uint8)
    prev = np.zeros(W, dtype=np.int64)
    for y in range(H):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int64)
        if ft == 0:
            cur = line
        elif ft == 1:                                    # Sub: running sum mod 256
            cur = np.cumsum(line) & 255
        elif ft == 2:                                    # Up
            cur = (line + prev) & 255
        elif ft in (3, 4):                               # Average / Paeth: sequential along the row
            cur = np.zeros(W, dtype=np.int64)
            a = c = 0
            for x in range(W):
                b = int(prev[x])
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                a = (int(line[x]) + pred) & 255
                cur[x] = a
                c = b
        else:
            raise ValueError(f"bad PNG filter type {ft}")
        out[y] = cur
        prev = cur
    return out


def encode_png_gray8(img):
    """(H, W) uint8 -> PNG bytes (filter 0); for fixtures and the synthetic example."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = np.concatenate([np.zeros((H, 1), dtype=np.uint8), img], axis=1).tobytes()
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def load_gray8(path):
    """PNG file -> (H, W) uint8 grayscale."""
    try:
        from PIL import Image
    except Exception:
        with open(path, "rb") as f:
            return decode_png_gray8(f.read())
    with Image.open(path) as im:
        return np.asarray(im.convert("L"), dtype=np.uint8)


class KittyDataset:
    """kitty.jl:29-73.  K: P0 as 4x4; Ti0 = inv(K) * P1 (camera 0 -> camera 1, entries below 1e-6 zeroed)."""

    def __init__(self, base_dir, sequence, stereo=True):
        frames_dir = os.path.join(base_dir, "sequences", sequence)
        with open(os.path.join(frames_dir, "calib.txt")) as f:
            Ks = f.read().splitlines()
        self.K = parse_matrix(Ks[0][4:])                 # "P0: ..." -> after the 4-character tag (Julia: [5:end])
        KT2 = parse_matrix(Ks[1][4:])
        Ti0 = np.linalg.inv(self.K) @ KT2
        Ti0[np.abs(Ti0) < 1e-6] = 0.0
        self.Ti0 = Ti0
        self.timestamps = read_timestamps(os.path.join(frames_dir, "times.txt"))
        self.left_frames_dir = os.path.join(frames_dir, "image_0")
        self.right_frames_dir = os.path.join(frames_dir, "image_1")
        poses_file = os.path.join(base_dir, "poses", sequence + ".txt")
        self.poses = read_poses(poses_file) if os.path.exists(poses_file) else []     # test sequences 11-21 ship none
        self.stereo = bool(stereo)

    def __len__(self):
        return len(self.poses) if self.poses else len(self.timestamps)

    def __getitem__(self, i):
        """Frame i (0-based; file %06d.png) -> (left, right) uint8 (H, W); right is left when not stereo (kitty.jl:90-101)."""
        if not 0 <= i < len(self):
            raise IndexError(i)
        left = load_gray8(os.path.join(self.left_frames_dir, f"{i:06d}.png"))
        right = load_gray8(os.path.join(self.right_frames_dir, f"{i:06d}.png")) if self.stereo else left
        return left, right

    @property
    def intrinsics(self):
        """(fx, fy, cx, cy) as main.jl:22-23 reads them."""
        return self.K[0, 0], self.K[1, 1], self.K[0, 2], self.K[1, 2]

    @property
    def baseline(self):
        """Stereo baseline in metres (|Ti0[0, 3]|)."""
        return abs(self.Ti0[0, 3])

    def get_camera_poses(self):
        """Positions and viewing directions of the ground-truth poses (kitty.jl:75-87)."""
        P = np.array(self.poses).reshape(-1, 4, 4)
        pos = P[:, :3, 3].copy()
        d = P[:, :3, 2].copy()
        return pos, d / np.linalg.norm(d, axis=1, keepdims=True)

    def __repr__(self):
        return f"Kitty Dataset:\n- Number of frames: {len(self)}\n- Intrinsics:\n{self.K}"


def write_synthetic_sequence(base_dir, sequence, left, right, cam, baseline, poses=None, dt=0.1):
    """A KITTI-odometry-shaped directory from float images in [0, 1] (H, W): used by tests and the example when no
    dataset is at hand (this image has no network).  Frames are quantised to 8 bits, like the real dataset."""
    fdir = os.path.join(base_dir, "sequences", sequence)
    os.makedirs(os.path.join(fdir, "image_0"), exist_ok=True)
    os.makedirs(os.path.join(fdir, "image_1"), exist_ok=True)
    os.makedirs(os.path.join(base_dir, "poses"), exist_ok=True)
    fx, fy, cx, cy = cam
    P0 = [fx, 0, cx, 0, 0, fy, cy, 0, 0, 0, 1, 0]
    P1 = [fx, 0, cx, -fx * baseline, 0, fy, cy, 0, 0, 0, 1, 0]
    with open(os.path.join(fdir, "calib.txt"), "w") as f:
        for k, P in enumerate((P0, P1, P0, P1)):
            f.write(f"P{k}: " + " ".join(f"{v:.12e}" for v in P) + "\n")
    with open(os.path.join(fdir, "times.txt"), "w") as f:
        for i in range(len(left)):
            f.write(f"{i * dt:.6e}\n")
    for i, (a, b) in enumerate(zip(left, right)):
        for d, im in (("image_0", a), ("image_1", b)):
            with open(os.path.join(fdir, d, f"{i:06d}.png"), "wb") as f:
                f.write(encode_png_gray8(np.round(np.clip(im, 0, 1) * 255).astype(np.uint8)))
    write_poses(os.path.join(base_dir, "poses", sequence + ".txt"), poses if poses is not None else [np.eye(4)] * len(left))
