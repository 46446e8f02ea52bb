"""CPU: known-answer tests of the ORB front-end restatement (oracle/orc_orb.cpp).  cv::ORB is upstream OpenCV --
parity with its binary is UNPINNED (no OpenCV, no fixtures in the reference); these tests pin the restatement to
the published definitions it claims to follow: the FAST-9 segment test and its score, OpenCV's per-level sizes
and feature quotas, the bilinear / Gaussian conventions on hand-checkable inputs, fastAtan2's documented 0.3
degree accuracy, and the steered-BRIEF bit layout."""
import math

import numpy as np


CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
          (-3, 1), (-2, 2), (-1, 3)]


def _ring(center, values):
    img = np.full((9, 9), center, np.uint8)
    for (dx, dy), v in zip(CIRCLE, values):
        img[4 + dy, 4 + dx] = v
    return img


def test_fast_segment_test_and_score(orc):
    # 9 contiguous brighter pixels by exactly 60, the rest equal to the centre: corner up to t = 59
    assert orc.orb_fast_score(_ring(100, [160] * 9 + [100] * 7), 4, 4, 20) == 59
    # only 8 contiguous: not a corner
    assert orc.orb_fast_score(_ring(100, [160] * 8 + [100] * 8), 4, 4, 20) == 0
    # arc wrapping around index 15 -> 0, darker side, weakest arc pixel decides the score
    vals = [40] * 4 + [100] * 7 + [40] * 4 + [55]
    assert orc.orb_fast_score(_ring(100, vals), 4, 4, 20) == 44
    # contrast exactly at the threshold is not enough (strict comparison), one more is
    assert orc.orb_fast_score(_ring(100, [120] * 9 + [100] * 7), 4, 4, 20) == 0
    assert orc.orb_fast_score(_ring(100, [121] * 9 + [100] * 7), 4, 4, 20) == 20
    # within 3 pixels of the border nothing is scored
    assert orc.orb_fast_score(_ring(100, [160] * 16), 2, 4, 20) == 0


def test_level_sizes_and_quotas(orc):
    lw, lh, sc = orc.orb_level_sizes(752, 480)
    assert lw.tolist() == [752, 627, 522, 435, 363, 302, 252, 210]
    assert lh.tolist() == [480, 400, 333, 278, 231, 193, 161, 134]
    assert abs(sc[3] - 1.2 ** 3) < 1e-6
    q = orc.orb_level_quota(1500)
    assert q.sum() == 1500 and np.all(np.diff(q) < 0)
    assert abs(q[0] / q[1] - 1.2) < 0.01
    assert orc.orb_level_quota(10).sum() == 10


def test_resize_and_blur_conventions(orc):
    const = np.full((37, 53), 77, np.uint8)
    assert np.all(orc.orb_resize(const, 44, 31) == 77)
    assert np.all(orc.orb_gauss7(const) == 77)
    # a horizontal ramp stays a ramp with slope src/dst under bilinear resampling (interior pixels)
    ramp = np.tile(np.arange(120, dtype=np.uint8), (20, 1))
    out = orc.orb_resize(ramp, 100, 20).astype(np.float64)
    exp = (np.arange(100) + 0.5) * 1.2 - 0.5
    assert np.abs(out[10, 2:-2] - exp[2:-2]).max() <= 0.51
    # the blur of an impulse is the outer product of the normalised 7-tap sigma-2 kernel (+- rounding)
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    k = np.exp(-np.arange(-3, 4) ** 2 / 8.0)
    k /= k.sum()
    assert np.abs(orc.orb_gauss7(imp)[7:14, 7:14] - 255 * np.outer(k, k)).max() <= 0.51
    # reflect-101 at the border: a column next to the edge sees its mirror image
    edge = np.zeros((9, 9), np.uint8)
    edge[:, 1] = 200
    b = orc.orb_gauss7(edge)
    assert abs(b[4, 0] - 200 * 2 * k[2]) <= 0.51   # column 1 is seen at offsets -1 (mirrored) and +1


def test_fast_atan2_accuracy(orc):
    rng = np.random.default_rng(0)
    for _ in range(500):
        y, x = rng.normal(size=2) * 1000
        ref = math.degrees(math.atan2(y, x)) % 360.0
        got = orc.orb_fast_atan2(float(y), float(x))
        assert min(abs(got - ref), 360 - abs(got - ref)) < 0.3
    assert orc.orb_fast_atan2(0.0, 1.0) == 0.0 and orc.orb_fast_atan2(1.0, 0.0) == 90.0


def test_descriptor_of_a_step_edge(orc):
    # left half dark, right half bright: the keypoints it yields (if any) must carry descriptors whose bits
    # follow the sign of the rotated test offsets; weak check: two identical images give identical output and
    # a flat image gives none
    img = np.zeros((200, 200), np.uint8)
    img[:, 100:] = 200
    img[60:140, 60:140] = np.random.default_rng(1).integers(0, 256, (80, 80))
    kp1, d1 = orc.orb_detect_describe(img, 500)
    kp2, d2 = orc.orb_detect_describe(img.copy(), 500)
    assert len(kp1) > 20 and np.array_equal(kp1, kp2) and np.array_equal(d1, d2)
    assert len(orc.orb_detect_describe(np.full((200, 200), 9, np.uint8), 500)[0]) == 0
    # keypoints respect the 19-pixel border at their own level and scores are >= the FAST threshold
    lw, lh, sc = orc.orb_level_sizes(200, 200)
    for x, y, ang, resp, octv in kp1:
        l = int(octv)
        xl, yl = x / sc[l], y / sc[l]
        assert 19 - 1e-3 <= xl <= lw[l] - 19 and 19 - 1e-3 <= yl <= lh[l] - 19
        assert resp >= 20 and 0 <= ang < 360
