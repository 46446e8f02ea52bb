"""CPU: hand-checkable known-answer tests that pin the oracle's restatement of the in-tree
reference code (include/visnav/keypoints.h, converter.h), plus the committed golden vectors."""
import math

import numpy as np
import pytest

from conftest import GOLDEN


def _desc_from_bits(bits):
    d = np.zeros(4, np.uint64)
    for b in bits:
        d[b // 64] |= np.uint64(1) << np.uint64(b % 64)
    return d


def test_converter_roundtrip_and_bit_order(orc):
    # the one assertion of the reference's own tests: to_bitset(to_opencv(d)) == d
    # (test/src/test_loop_closure_utils.cpp:152-159), plus the MSB-first rule of converter.h:27
    rng = np.random.default_rng(0)
    d = rng.integers(0, 2 ** 63, size=(50, 4), dtype=np.int64).astype(np.uint64)
    b = orc.bitset_to_bytes(d)
    assert np.array_equal(orc.bytes_to_bitset(b), d)
    one = _desc_from_bits([0])           # bit 0 -> byte 0, bit 7
    assert orc.bitset_to_bytes(one)[0, 0] == 0x80 and orc.bitset_to_bytes(one)[0, 1:].sum() == 0
    b9 = _desc_from_bits([9])            # bit 9 -> byte 1, bit 6
    assert orc.bitset_to_bytes(b9)[0, 1] == 0x40
    b255 = _desc_from_bits([255])        # bit 255 -> byte 31, bit 0
    assert orc.bitset_to_bytes(b255)[0, 31] == 0x01


def test_matcher_micro_kats(orc):
    z = _desc_from_bits([])
    # identical single descriptors: best 0 < 70, second stays 256 -> match
    assert orc.match_descriptors([z], [z]).tolist() == [[0, 0]]
    # threshold edge: distance 69 matches, 70 does not (best1_d >= threshold, keypoints.h:355)
    d69, d70 = _desc_from_bits(range(69)), _desc_from_bits(range(70))
    assert orc.match_descriptors([z], [d69]).tolist() == [[0, 0]]
    assert orc.match_descriptors([z], [d70]).tolist() == []
    # ratio edge: best 10, second 12 -> 12 < 10*1.2 is false -> accepted; second 11 -> rejected
    a10, a11, a12 = _desc_from_bits(range(10)), _desc_from_bits(range(100, 111)), _desc_from_bits(range(100, 112))
    assert orc.match_descriptors([z], [a10, a12]).tolist() == [[0, 0]]
    assert orc.match_descriptors([z], [a10, a11]).tolist() == []
    # ties: two columns at the same distance -> second == best -> ratio test rejects (d < d*1.2)
    t1, t2 = _desc_from_bits([1, 2, 3]), _desc_from_bits([4, 5, 6])
    assert orc.match_descriptors([z], [t1, t2]).tolist() == []
    # ... unless the distance is 0 (0 < 0*1.2 is false): the LOWEST index wins the tie
    assert orc.match_descriptors([z], [z, z]).tolist() == [[0, 0]]
    # failed cross-check: row 0's best is column 0, but column 0 prefers row 1
    r0, r1 = _desc_from_bits(range(20)), _desc_from_bits([])
    c0 = _desc_from_bits([])
    assert orc.match_descriptors([r0, r1], [c0]).tolist() == [[1, 0]]
    # empty inputs
    assert len(orc.match_descriptors(np.zeros((0, 4), np.uint64), [z])) == 0
    assert len(orc.match_descriptors([z], np.zeros((0, 4), np.uint64))) == 0


def test_angle_of_half_planes(orc):
    # bright right half -> centroid on +x -> angle 0; bright bottom -> +y -> pi/2; left -> pi; top -> -pi/2
    h = w = 64
    c = np.array([[32.0, 32.0]])
    yy, xx = np.mgrid[0:h, 0:w]
    for img, expect in ((xx > 32, 0.0), (yy > 32, math.pi / 2), (xx < 32, math.pi), (yy < 32, -math.pi / 2)):
        a = orc.compute_angles((img * 200).astype(np.uint8), c, True)[0]
        assert a == pytest.approx(expect, abs=1e-12)
    assert orc.compute_angles((xx > 32).astype(np.uint8) * 200, c, False)[0] == 0.0
    # exact integer moments of a single bright pixel at offset (+3, -2): m10 = 3*v, m01 = -2*v
    img = np.zeros((h, w), np.uint8)
    img[30, 35] = 100
    m01, m10 = orc.patch_moments(img, c)
    assert (m01[0], m10[0]) == (-200, 300)
    assert orc.compute_angles(img, c, True)[0] == math.atan2(-200.0, 300.0)


def test_descriptor_of_constant_image_is_zero(orc):
    img = np.full((64, 64), 77, np.uint8)  # '<' is strict (keypoints.h:215)
    d = orc.compute_descriptors(img, np.array([[32.0, 32.0]]), np.array([0.3]))
    assert not d.any()


def test_descriptor_first_bit_unrotated(orc):
    # pattern row 0 is (8,-3) vs (9,5) (keypoints.h:56,76,96,114): bit 0 = I(cx+8, cy-3) < I(cx+9, cy+5)
    img = np.full((64, 64), 50, np.uint8)
    img[32 + 5, 32 + 9] = 60
    d = orc.compute_descriptors(img, np.array([[32.0, 32.0]]), np.array([0.0]))
    assert int(d[0, 0]) & 1 == 1
    img[32 - 3, 32 + 8] = 70
    d = orc.compute_descriptors(img, np.array([[32.0, 32.0]]), np.array([0.0]))
    assert int(d[0, 0]) & 1 == 0


def test_good_features_semantics(orc):
    # an isolated bright square gives 4 corner responses; flat image gives none
    img = np.full((120, 160), 30, np.uint8)
    assert len(orc.good_features(img, 100)) == 0
    img[40:80, 60:100] = 220
    pts = orc.good_features(img, 100)
    assert len(pts) >= 4
    # min distance 8 between any two accepted corners
    d = pts[:, None, :] - pts[None, :, :]
    d2 = (d ** 2).sum(-1) + np.eye(len(pts), dtype=np.int64) * 10 ** 6
    assert d2.min() >= 64
    # response-descending order
    r = orc.min_eig_response(img)
    vals = r[pts[:, 1], pts[:, 0]]
    assert np.all(np.diff(vals) <= 0)
    # detectKeypoints drops points closer than 19 px to the border (keypoints.h:145-149)
    img2 = np.full((120, 160), 30, np.uint8)
    img2[5:30, 5:30] = 220
    kp = orc.detect_keypoints(img2, 100)
    assert np.all(kp[:, 0] >= 19) and np.all(kp[:, 1] >= 19)


@pytest.mark.parametrize("k", range(16))  # 16 real EuRoC pairs of the reference's data/euroc_V1 (tools/make_golden.py)
def test_oracle_reproduces_golden(orc, k):
    g = np.load(GOLDEN / ("euroc_pair%d.npz" % k))
    descs = []
    for c in (0, 1):
        img = g["img%d" % c]
        xy, ang, desc = orc.detect_describe(img, 1500, True)
        assert np.array_equal(xy.astype(np.int32), g["xy%d" % c])
        assert np.array_equal(ang.view(np.uint64), g["angle_bits%d" % c])
        assert np.array_equal(desc, g["desc%d" % c])
        resp = orc.min_eig_response(img)
        assert np.bitwise_xor.reduce(resp.view(np.uint32).ravel()) == g["resp_bits_xor%d" % c]
        descs.append(desc)
    assert np.array_equal(orc.match_descriptors(descs[0], descs[1], 70, 1.2), g["matches"])
