"""Params / Camera mirrors (reference: src/params.jl:58-82, src/camera.jl:1-67).
Only the fields the hot path reads."""
from dataclasses import dataclass


@dataclass
class Params:
    stereo: bool = False
    max_nb_keypoints: int = 1000
    max_distance: int = 35
    max_ktl_distance: float = 1.0
    pyramid_levels: int = 3
    pyramid_sigma: float = 1.0
    window_size: int = 9
    max_reprojection_error: float = 3.0
    min_cov_score: int = 25
    do_local_matching: bool = False
    do_local_bundle_adjustment: bool = True


@dataclass
class Camera:
    fx: float
    fy: float
    cx: float
    cy: float
    height: int
    width: int

    def project(self, point):
        """camera.jl:62-67: (x, y, z) -> (y, x) pixel."""
        inv_z = 1.0 / point[2]
        return (self.fy * point[1] * inv_z + self.cy, self.fx * point[0] * inv_z + self.cx)

    @property
    def intrinsics(self):
        return (self.fx, self.fy, self.cx, self.cy)
