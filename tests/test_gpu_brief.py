"""GPU parity: slam_describe vs the CPU oracle (extractor.jl:103-105) -- bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_describe_bit_exact_and_border_drop(slam, orc, texture):
    H, W = 120, 160
    img = texture(H, W)[0][0]
    kp = orc.detect(img, np.zeros((0, 2)), max_points=150)
    kp = np.concatenate([kp, [[1, 1], [5, 80], [6, 80], [H - 5, W - 5], [H - 4, 20], [60, W]]])
    pat = slam.brief_pattern()
    bits, rc = slam.describe(slam.Extractor(150, 17, (4, 5), 35), img, kp, pattern=pat)
    rbits, rrc = orc.describe(img, kp, pat)
    assert np.array_equal(rc, rrc) and np.array_equal(bits, rbits)
    assert len(rc) < len(kp) and bits.shape[1] == 4
    # hamming distance to itself is 0, to a shifted keypoint's descriptor is not
    assert (bits[0] ^ bits[0]).sum() == 0 and len(np.unique(bits, axis=0)) > 0.9 * len(bits)
