"""GPU parity: project_landmarks / find_matches_landmarks (visual-slam_amd/csrc/vo.hip) vs the oracle.
fp64 projection in the oracle's operation order: bit-exact for ds / pinhole / eucm (kb4 goes through
atan2: device and glibc differ in the last bits -> same kept set, 1e-12 px); matches: identical pairs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INTR = {0: [351.0, 350.0, 365.9, 249.3, -0.2385, 0.5679, 0, 0], 1: [351.0, 350.0, 365.9, 249.3, 0, 0, 0, 0],
        2: [351.0, 350.0, 365.9, 249.3, 0.6, 1.1, 0, 0], 3: [351.0, 350.0, 365.9, 249.3, 0.01, -0.004, 0.002, -0.0005]}


def _scene(synth, seed, n_lms=3000):
    rng = np.random.default_rng(seed)
    pose = np.concatenate([synth.axis_angle_q(rng.normal(size=3), 0.3), rng.normal(0, 0.5, 3)])
    R = synth.quat_R(pose[:4])
    pc = np.stack([rng.uniform(-8, 8, n_lms), rng.uniform(-5, 5, n_lms), rng.uniform(-2, 12, n_lms)], -1)
    pw = pc @ R.T + pose[4:]
    return pose, pw


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_project_landmarks(ctx, orc, synth, model):
    pose, pw = _scene(synth, model)
    uv, idx = ctx.project_landmarks(pose, model, INTR[model], 752, 480, pw, 0.1)
    euv, eidx = orc.project_landmarks(pose, model, INTR[model], 752, 480, pw, 0.1)
    assert np.array_equal(idx, eidx) and 200 < len(idx) < len(pw)
    if model == 3:
        assert np.allclose(uv, euv, rtol=0, atol=1e-12)
    else:
        assert np.array_equal(uv.view(np.uint64), euv.view(np.uint64))


def test_project_landmarks_edges(ctx, orc):
    ident = [0, 0, 0, 1, 0, 0, 0]
    pin = [100.0, 100.0, 50.0, 40.0, 0, 0, 0, 0]
    pts = np.array([[0, 0, 1.0], [0, 0, 0.05], [0.5, 0, 1.0], [0.51, 0, 1.0], [-0.51, 0, 1.0], [0, 0.4, 1.0], [0, 0, -1.0]])
    uv, idx = ctx.project_landmarks(ident, 1, pin, 100, 80, pts, 0.1)
    assert idx.tolist() == [0, 2, 5] and uv.tolist() == [[50.0, 40.0], [100.0, 40.0], [50.0, 80.0]]
    uv, idx = ctx.project_landmarks(ident, 1, pin, 100, 80, np.zeros((0, 3)), 0.1)
    assert len(idx) == 0


def _match_case(synth, seed, n_kp, n_lms, max_obs, planted=0.6):
    rng = np.random.default_rng(seed)
    kp_xy = np.stack([rng.integers(19, 733, n_kp), rng.integers(19, 461, n_kp)], -1).astype(np.float64)
    kp_desc = synth.random_descriptors(rng, n_kp)
    n_obs = rng.integers(0, max_obs + 1, n_lms)
    start = np.concatenate([[0], np.cumsum(n_obs)]).astype(np.int32)
    obs = synth.random_descriptors(rng, int(start[-1]))
    proj_uv = np.stack([rng.uniform(0, 752, n_lms), rng.uniform(0, 480, n_lms)], -1)
    # plant: landmark l projects near keypoint k and one of its observations is a noisy copy of k's descriptor
    for l in range(n_lms):
        if n_obs[l] and rng.random() < planted:
            k = int(rng.integers(n_kp))
            proj_uv[l] = kp_xy[k] + rng.uniform(-12, 12, 2)
            o = int(start[l] + rng.integers(n_obs[l]))
            obs[o] = synth.flip_bits(rng, kp_desc[k:k + 1], int(rng.choice([0, 0, 3, 20, 50, 69, 70, 75])))[0]
    proj_lm = rng.permutation(n_lms).astype(np.int32)  # projected order != landmark order
    return kp_xy, kp_desc, proj_uv[proj_lm], proj_lm, start, obs


@pytest.mark.parametrize("seed,n_kp,n_lms,max_obs", [(1, 1, 1, 3), (2, 1300, 2500, 20), (3, 257, 4000, 70), (4, 1500, 63, 5),
                                                      (5, 64, 300, 200)])
def test_find_matches_landmarks(ctx, orc, synth, seed, n_kp, n_lms, max_obs):
    c = _match_case(synth, seed, n_kp, n_lms, max_obs)
    got = ctx.find_matches_landmarks(*c, 20.0, 70, 1.2)
    exp = orc.find_matches_landmarks(*c, 20.0, 70, 1.2)
    assert np.array_equal(got, exp)
    if n_kp > 100 and n_lms > 1000:
        assert len(got) > 50
    for r, t, q in ((5.0, 70, 1.2), (40.0, 100, 1.0), (20.0, 1, 3.0)):
        assert np.array_equal(ctx.find_matches_landmarks(*c, r, t, q), orc.find_matches_landmarks(*c, r, t, q))


def test_find_matches_tie_semantics(ctx, orc):
    # the partial_sort tie cases of tests/test_oracle_vo.py, through the kernel's state machine
    z = np.zeros(4, np.uint64)

    def d(n):
        o = np.zeros(4, np.uint64)
        for b in range(n):
            o[b // 64] |= np.uint64(1) << np.uint64(b % 64)
        return o

    for dists in ([0, 0], [0, 7, 0], [7, 0, 0], [0, 0, 0], [5, 0, 9, 0, 3], [3, 3], [4, 2, 2, 9, 2], [9, 8, 7, 6, 5, 5, 6, 5]):
        proj = [[100.0 + 0.1 * i, 100.0] for i in range(len(dists))]
        start = np.arange(len(dists) + 1, dtype=np.int32)
        obs = np.stack([d(v) for v in dists])
        args = ([[100.0, 100.0]], [z], proj, np.arange(len(dists), dtype=np.int32), start, obs, 20.0, 70, 1.0)
        assert np.array_equal(ctx.find_matches_landmarks(*args), orc.find_matches_landmarks(*args)), dists


def test_find_matches_radius_boundary(ctx, orc):
    # the kernel tests dx^2 + dy^2 < T with the host-computed T that makes it the same decision as the reference's
    # (p_2d - kp).norm() < match_max_dist_2d (vo_utils.h:108) for EVERY double: projected points exactly on the circle
    # (3-4-5 triangles: norm == radius, not a hit), one ulp inside / outside it, squared sums within an ulp of radius^2
    # (where sqrt rounds to the radius itself), radii that are not representable squares
    z = np.zeros(4, np.uint64)
    for radius in (20.0, 5.0, 0.1, 19.999999999999996, 7.3, 1e-3, 123.456):
        pts = []
        for ang in np.linspace(0.0, 2 * np.pi, 97):
            for scale in (1.0, np.nextafter(1.0, 0), np.nextafter(1.0, 2), 1 - 1e-15, 1 + 1e-15, 1 - 3e-16, 1 + 3e-16):
                pts.append([300.0 + radius * scale * np.cos(ang), 200.0 + radius * scale * np.sin(ang)])
        if radius == 20.0:
            pts += [[312.0, 216.0], [288.0, 184.0], [312.0, np.nextafter(216.0, 0)], [312.0, np.nextafter(216.0, 1e9)]]
        n = len(pts)
        start = np.arange(n + 1, dtype=np.int32)
        obs = np.zeros((n, 4), np.uint64)
        # every landmark alone against the keypoint: n single-landmark calls would be slow -- instead give landmark i the
        # descriptor distance i % 60 and compare the whole result with the oracle for several thresholds
        for i in range(n):
            for b in range(i % 60):
                obs[i, b // 64] |= np.uint64(1) << np.uint64(b % 64)
        for sub in (slice(0, n), slice(n // 3, n), slice(0, n, 7), slice(5, n, 11)):
            idx = np.arange(n, dtype=np.int32)[sub]
            args = ([[300.0, 200.0]], [z], np.asarray(pts)[sub], np.arange(len(idx), dtype=np.int32),
                    np.arange(len(idx) + 1, dtype=np.int32), obs[sub], radius, 70, 1.0)
            assert np.array_equal(ctx.find_matches_landmarks(*args), orc.find_matches_landmarks(*args)), (radius, sub)
        # and landmark by landmark on the circle itself (the first 97 * 7 points): hit / no hit must agree one by one
        for i in list(range(0, 97 * 7, 13)) + list(range(97 * 7, n)):
            args = ([[300.0, 200.0]], [z], [pts[i]], np.zeros(1, np.int32), np.array([0, 1], np.int32), obs[:1], radius, 70, 1.0)
            assert np.array_equal(ctx.find_matches_landmarks(*args), orc.find_matches_landmarks(*args)), (radius, i)


def test_find_matches_empty(ctx):
    z = np.zeros((0, 4), np.uint64)
    assert len(ctx.find_matches_landmarks(np.zeros((0, 2)), z, np.zeros((0, 2)), np.zeros(0, np.int32), np.zeros(1, np.int32), z)) == 0
    assert len(ctx.find_matches_landmarks([[50.0, 50.0]], np.zeros((1, 4), np.uint64), np.zeros((0, 2)), np.zeros(0, np.int32),
                                          np.zeros(1, np.int32), z)) == 0


def _frame_and_map_inputs(vsl, ctx, synth, seed, n_lms):
    """A real frame in a frame-store slot + landmarks whose observation descriptors are partly noisy copies of
    that frame's keypoint descriptors (so that matches exist), partly random."""
    rng = np.random.default_rng(seed)
    left, _ = synth.stereo_pair(seed)
    fr = vsl.Frames(ctx, 2, 752, 480, 1500, max_pairs=1)
    fr.upload(0, left)
    fr.detect_describe(0, 1, 1500, True)
    kp_xy, _, kp_desc = fr.keypoints(0)
    # landmarks: points along the rays of random pixels at random depth, seen from a slightly moved pose
    pose = np.array([0.01, -0.02, 0.005, 1.0, 0.03, -0.02, 0.01])
    pose[:4] /= np.linalg.norm(pose[:4])
    u = rng.uniform(-50, 802, n_lms)
    v = rng.uniform(-50, 530, n_lms)
    z = rng.uniform(-1.0, 8.0, n_lms)   # some behind the camera
    fx, fy, cx, cy = INTR[1][:4]
    pts_c = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    q = pose[:4]
    R = np.array([[1 - 2 * (q[1] ** 2 + q[2] ** 2), 2 * (q[0] * q[1] - q[2] * q[3]), 2 * (q[0] * q[2] + q[1] * q[3])],
                  [2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[0] ** 2 + q[2] ** 2), 2 * (q[1] * q[2] - q[0] * q[3])],
                  [2 * (q[0] * q[2] - q[1] * q[3]), 2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[0] ** 2 + q[1] ** 2)]])
    points = pts_c @ R.T + pose[4:]
    # observations
    start, pool = [0], []
    for j in range(n_lms):
        k = int(rng.integers(0, 6))
        for _ in range(k):
            if len(kp_desc) and rng.random() < 0.6:
                # a keypoint near the (pinhole) projection, with a few flipped bits
                d2 = (kp_xy[:, 0] - u[j]) ** 2 + (kp_xy[:, 1] - v[j]) ** 2
                src = kp_desc[int(np.argmin(d2))].copy()
                for b in rng.integers(0, 256, int(rng.integers(0, 30))):
                    src[b // 64] ^= np.uint64(1) << np.uint64(b % 64)
                pool.append(src)
            else:
                pool.append(rng.integers(0, 2 ** 63, 4).astype(np.uint64))
        start.append(len(pool))
    pool = np.array(pool, np.uint64).reshape(-1, 4)
    return fr, kp_xy, kp_desc, pose, points, np.array(start, np.int32), pool


@pytest.mark.parametrize("seed,n_lms", [(3, 1), (4, 700), (5, 5000)])
def test_map_track_equals_host_buffer_path(vsl, ctx, synth, seed, n_lms):
    fr, kp_xy, kp_desc, pose, points, start, pool = _frame_and_map_inputs(vsl, ctx, synth, seed, n_lms)
    model, intr = 1, INTR[1]
    # host-buffer path (each step verified against the oracle by the tests above)
    uv, idx = ctx.project_landmarks(pose, model, intr, 752, 480, points, 0.1)
    exp = ctx.find_matches_landmarks(kp_xy, kp_desc, uv, idx, start, pool, 20.0, 70, 1.2)
    # device-resident path
    m = vsl.Map(ctx, 16, 16)   # tiny capacities: growth is part of the test
    first = m.append_descriptors(pool[:len(pool) // 2])
    assert first == 0
    assert m.append_descriptors(pool[len(pool) // 2:]) == len(pool) // 2
    m.set_landmarks(points, start, np.arange(len(pool), dtype=np.int32))
    assert m.info() == (n_lms, len(pool), len(pool))
    got, n_proj = m.track(fr, 0, pose, model, intr, 752, 480, 0.1, 20.0, 70, 1.2)
    assert n_proj == len(uv)
    assert np.array_equal(got, exp)
    if n_lms >= 700:
        assert len(exp) > 20   # the fixture does produce matches
    # tracking twice gives the same answer (scratch state is per call)
    got2, _ = m.track(fr, 0, pose, model, intr, 752, 480, 0.1, 20.0, 70, 1.2)
    assert np.array_equal(got2, exp)
    # ... and the keypoint positions can ride along in the same round trip (vsl_map_track_corners)
    got3, n_proj3, xy3 = m.track(fr, 0, pose, model, intr, 752, 480, 0.1, 20.0, 70, 1.2, with_corners=True)
    assert np.array_equal(got3, exp) and n_proj3 == n_proj and np.array_equal(xy3, fr.keypoints(0)[0])
    m.close()
    fr.close()


def test_map_track_resolves_pending_ties_and_large_maps(vsl, ctx, orc, synth):
    # (1) the rBRIEF near-tie guard rides along with the track results: with the guard band widened so that ties ARE
    # pending, the call must patch the descriptors (host libm) and match once more -- same matches as with the ties
    # resolved up front; (2) a map of > 1024 landmarks spans several workgroups of the projection kernel: the stream-scan
    # compaction must keep the landmark order (40,000 landmarks = 40 chained workgroups), call after call (the chain
    # words are epoch-tagged, never reset)
    fr, kp_xy, kp_desc, pose, points, start, pool = _frame_and_map_inputs(vsl, ctx, synth, 6, 40000)
    model, intr = 1, INTR[1]
    m = vsl.Map(ctx)
    m.append_descriptors(pool)
    m.set_landmarks(points, start, np.arange(len(pool), dtype=np.int32))
    uv, idx = ctx.project_landmarks(pose, model, intr, 752, 480, points, 0.1)
    exp = ctx.find_matches_landmarks(kp_xy, kp_desc, uv, idx, start, pool, 20.0, 70, 1.2)
    for _ in range(3):
        got, n_proj = m.track(fr, 0, pose, model, intr, 752, 480, 0.1, 20.0, 70, 1.2)
        assert n_proj == len(uv) and np.array_equal(got, exp)
    # chain positions drawn from the atomic ticket (the rule above 256 workgroups = 262 k landmarks, forced here at
    # 40 workgroups): same ordered output, call after call (the ticket word resets itself)
    ctx.set_diagnostic("vo_chain_ticket", 1)
    try:
        for _ in range(3):
            got, n_proj = m.track(fr, 0, pose, model, intr, 752, 480, 0.1, 20.0, 70, 1.2)
            assert n_proj == len(uv) and np.array_equal(got, exp)
    finally:
        ctx.set_diagnostic("vo_chain_ticket", 0)
    left = synth.stereo_pair(6)[0]
    ctx.set_tie_eps(1e-3)   # ~1e-3 of the samples become "near ties": pending when track is called
    try:
        fr.detect_describe(0, 1, 1500, True)
        got, _ = m.track(fr, 0, pose, model, intr, 752, 480, 0.1, 20.0, 70, 1.2)
    finally:
        ctx.set_tie_eps(1e-12)
    assert np.array_equal(fr.keypoints(0)[2], orc.detect_describe(left, 1500, True)[2])   # the ties were resolved
    assert np.array_equal(got, exp)
    m.close()
    fr.close()


def test_map_descriptors_copied_from_a_frame_slot(vsl, ctx, synth):
    fr, kp_xy, kp_desc, pose, points, start, pool = _frame_and_map_inputs(vsl, ctx, synth, 9, 50)
    m = vsl.Map(ctx)
    ids = np.array([5, 0, 17, len(kp_desc) - 1, 5], np.int32)
    base = m.append_descriptors(pool)                      # host descriptors first
    first = m.append_descriptors_from_frame(fr, 0, ids)    # then device-to-device copies
    assert base == 0 and first == len(pool)
    # landmarks placed exactly on those keypoints (pinhole rays at depth 2), one copied observation each:
    # every one of them must match its own keypoint at distance 0
    fx, fy, cx, cy = INTR[1][:4]
    uvk = kp_xy[ids[:4]]
    pts = np.stack([(uvk[:, 0] - cx) / fx * 2.0, (uvk[:, 1] - cy) / fy * 2.0, np.full(4, 2.0)], 1)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
    m.set_landmarks(pts, np.arange(5, dtype=np.int32), first + np.arange(4, dtype=np.int32))
    got, n_proj = m.track(fr, 0, ident, 1, INTR[1], 752, 480, 0.1, 3.0, 70, 1.2)
    assert n_proj == 4
    assert sorted(map(tuple, got.tolist())) == sorted((int(ids[k]), k) for k in range(4))
    with pytest.raises(Exception):
        m.set_landmarks(pts, np.arange(5, dtype=np.int32), np.array([0, 1, 2, 10 ** 6], np.int32))
    with pytest.raises(Exception):
        m.append_descriptors_from_frame(fr, 0, np.array([99999], np.int32))
    m.close()
    fr.close()
