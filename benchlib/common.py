"""bench.py's shared constants and byte counts."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KF_EVERY = 5
RIGHT_TARGET_ONLY = os.environ.get("SLAM_BENCH_RIGHT_FULL") is None     # right frames are only matched INTO (mapper.jl:51-66): layers only above level 0 (SLAM_PYR_TARGET_ONLY)
CULL_FRACTION = 0.15              # share of tracked keypoints the map drops per key-frame
N_FRAMES = 8                      # distinct rendered frames, played ping-pong
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0       # same guide: measured-achievable stream rate (SURVEY 8d's denominator, quoted beside the spec)


def frame_sequence(n_steps):
    fwd = list(range(N_FRAMES)) + list(range(N_FRAMES - 2, 0, -1))
    return [fwd[i % len(fwd)] for i in range(n_steps + 1)]



def pyramid_bytes(H, W, levels):
    """per level read the layer + write layer, Iy, Ix, Iyy, Ixx, Iyx = 7 * 8 * sum(H_l W_l)."""
    tot = 0
    for _ in range(levels + 1):
        tot += H * W; H = (H + 1) // 2; W = (W + 1) // 2
    return 7 * 8 * tot


def iir_rows_bytes(H, W, levels):
    """k_iir_rows algorithmic bytes per pyramid: every element of each plane it filters read once + written once."""
    tot = 0
    for l in range(levels + 1):
        planes = 4 if l < levels else 3
        tot += planes * H * W * 16; H = (H + 1) // 2; W = (W + 1) // 2
    return tot


def frame_sequence_n(n_frames, n_steps):
    fwd = list(range(n_frames)) + list(range(n_frames - 2, 0, -1))
    return [fwd[i % len(fwd)] for i in range(n_steps + 1)]


