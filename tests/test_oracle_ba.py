"""CPU: known-answer / property tests that pin the oracle's BA restatement (camera_models.h,
reprojection.h:81-105, local_parameterization_se3.hpp) and its [upstream] Ceres-style LM loop."""
import numpy as np
import pytest


def _arr(orc, d):
    return orc.BaArrays(d["poses"], d["cam_fixed"], d["cam_intr"], d["intr"], d["points"], d["obs_cam"],
                        d["obs_lm"], d["obs_uv"], d["cam_model"])


INTR = {0: [350.0, 348.0, 365.0, 249.0, -0.24, 0.57, 0, 0],      # ds
        1: [350.0, 348.0, 365.0, 249.0, 0, 0, 0, 0],              # pinhole
        2: [350.0, 348.0, 365.0, 249.0, 0.6, 1.1, 0, 0],          # eucm
        3: [350.0, 348.0, 365.0, 249.0, 0.01, -0.004, 0.002, -0.0005]}  # kb4


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_project_optical_axis_hits_principal_point(orc, model):
    uv = orc.project(model, INTR[model], [0.0, 0.0, 1.0])
    assert uv == pytest.approx([365.0, 249.0], abs=1e-12)


def test_project_known_values(orc):
    # pinhole: u = fx x/z + cx (camera_models.h:88-89)
    assert orc.project(1, INTR[1], [0.2, -0.1, 2.0]) == pytest.approx([350 * 0.1 + 365, 348 * -0.05 + 249])
    # double sphere with xi = 0, alpha = 0 degenerates to pinhole (camera_models.h:259-266)
    p = [0.3, 0.2, 1.5]
    assert orc.project(0, [350, 348, 365, 249, 0, 0, 0, 0], p) == pytest.approx(orc.project(1, INTR[1], p))
    # eucm with alpha = 0 is pinhole too
    assert orc.project(2, [350, 348, 365, 249, 0, 1, 0, 0], p) == pytest.approx(orc.project(1, INTR[1], p))
    # kb4 with zero distortion: u = fx * theta * x / r + cx
    th = np.arctan2(np.hypot(0.3, 0.2), 1.5)
    assert orc.project(3, [350, 348, 365, 249, 0, 0, 0, 0], p) == pytest.approx(
        [350 * th * 0.3 / np.hypot(0.3, 0.2) + 365, 348 * th * 0.2 / np.hypot(0.3, 0.2) + 249])


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_jacobian_against_finite_differences(orc, synth, model):
    rng = np.random.default_rng(model)
    pose = np.concatenate([synth.axis_angle_q(rng.normal(size=3), 0.4), rng.normal(size=3)])
    pw = pose[4:] + synth.quat_R(pose[:4]) @ np.array([0.4, -0.3, 3.0])
    uv = np.array([400.0, 260.0])
    r, Jp, Jl = orc.ba_residual_jacobian(model, pose, pw, INTR[model], uv)
    assert np.allclose(r, orc.ba_residual(model, pose, pw, INTR[model], uv), atol=1e-13)
    h = 1e-6
    for j in range(6):
        d = np.zeros(6); d[j] = h
        rp = orc.ba_residual(model, orc.se3_plus(pose, d), pw, INTR[model], uv)
        rm = orc.ba_residual(model, orc.se3_plus(pose, -d), pw, INTR[model], uv)
        assert (rp - rm) / (2 * h) == pytest.approx(Jp[:, j], rel=1e-6, abs=1e-6)
    for j in range(3):
        d = np.zeros(3); d[j] = h
        rp = orc.ba_residual(model, pose, pw + d, INTR[model], uv)
        rm = orc.ba_residual(model, pose, pw - d, INTR[model], uv)
        assert (rp - rm) / (2 * h) == pytest.approx(Jl[:, j], rel=1e-6, abs=1e-6)


def test_se3_plus_properties(orc, synth):
    pose = np.concatenate([synth.axis_angle_q([1, 2, 3], 0.7), [0.1, -0.2, 0.3]])
    assert orc.se3_plus(pose, np.zeros(6)) == pytest.approx(pose, abs=1e-15)
    # pure translation in the body frame: t' = t + R v, q unchanged (T * exp(delta))
    v = np.array([0.1, 0.2, -0.05])
    out = orc.se3_plus(pose, np.concatenate([v, np.zeros(3)]))
    assert out[:4] == pytest.approx(pose[:4]) and out[4:] == pytest.approx(pose[4:] + synth.quat_R(pose[:4]) @ v)
    # pure rotation about z by 0.3 rad: q' = q * (0, 0, sin .15, cos .15)
    out = orc.se3_plus(pose, [0, 0, 0, 0, 0, 0.3])
    assert out[:4] == pytest.approx(synth.quat_mul(pose[:4], [0, 0, np.sin(0.15), np.cos(0.15)]))
    assert np.linalg.norm(out[:4]) == pytest.approx(1.0, abs=1e-15)


def test_bundle_adjust_recovers_noise_free_problem(orc, synth):
    d = synth.ba_problem(3, n_kf=4, n_lms=300, pix_noise=0.0, outlier_frac=0.0, integer_pixels=False)
    arr = _arr(orc, d)
    s = orc.bundle_adjust(arr, max_iters=50)
    assert s.final_cost < 1e-9 * s.initial_cost
    assert np.allclose(arr.points, d["gt_points"], atol=1e-5)
    assert s.iterations <= 50 and s.successful_steps >= 3


def test_bundle_adjust_policy(orc, synth):
    d = synth.ba_problem(4, n_kf=5, n_lms=800)
    arr = _arr(orc, d)
    p0 = arr.poses.copy()
    s = orc.bundle_adjust(arr, max_iters=20)
    assert s.final_cost < s.initial_cost
    assert s.iterations <= 20
    # fixed cameras (SetParameterBlockConstant, map_utils.h:364-366) never move
    fixed = d["cam_fixed"].astype(bool)
    assert np.array_equal(arr.poses[fixed], p0[fixed])
    assert np.allclose(np.linalg.norm(arr.poses[:, :4], axis=1), 1.0, atol=1e-12)
    # max_num_iterations is honoured (map_utils.h:407)
    arr2 = _arr(orc, d)
    s2 = orc.bundle_adjust(arr2, max_iters=2)
    assert s2.iterations == 2 and s2.termination == 0
    # threads only change the summation order of the cost partials
    arr3 = _arr(orc, d)
    s3 = orc.bundle_adjust(arr3, max_iters=20, threads=4)
    assert s3.final_cost == pytest.approx(s.final_cost, rel=1e-9)


def test_linearize_partition_is_additive(orc, synth):
    # the multi-GPU global-BA path relies on S, g, cost being sums over landmark ranges
    d = synth.ba_problem(5, n_kf=4, n_lms=500)
    arr = _arr(orc, d)
    S, g, c = orc.ba_linearize(arr)
    L = len(arr.points)
    S1, g1, c1 = orc.ba_linearize(arr, lm_first=0, lm_count=L // 3)
    S2, g2, c2 = orc.ba_linearize(arr, lm_first=L // 3, lm_count=L - L // 3)
    assert np.allclose(S1 + S2, S, rtol=1e-12, atol=1e-9)
    assert np.allclose(g1 + g2, g, rtol=1e-12, atol=1e-9)
    assert c1 + c2 == pytest.approx(c, rel=1e-12)
    assert np.allclose(S, S.T, rtol=1e-9, atol=1e-9 * np.abs(S).max())


# ---- optimize_intrinsics = true (map_utils.h:324, :397-403)
INTR8 = {0: [350.0, 348.0, 365.0, 249.0, -0.24, 0.57, 0, 0],
         1: [350.0, 348.0, 365.0, 249.0, 0, 0, 0, 0],
         2: [350.0, 348.0, 365.0, 249.0, 0.6, 1.1, 0, 0],
         3: [350.0, 348.0, 365.0, 249.0, 0.01, -0.004, 0.002, -0.0005]}
N_INTR = {0: 6, 1: 4, 2: 6, 3: 8}


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_intrinsics_jacobian_against_finite_differences(orc, synth, model):
    d = synth.ba_problem(70 + model, n_kf=2, n_lms=60)
    intr = np.array(INTR8[model])
    rng = np.random.default_rng(model)
    for i in rng.choice(len(d["obs_cam"]), 12, replace=False):
        pose, pt, uv = d["poses"][d["obs_cam"][i]], d["points"][d["obs_lm"][i]], d["obs_uv"][i]
        Ji = orc.ba_residual_jacobian_intr(model, pose, pt, intr, uv)
        for j in range(8):
            h = 1e-6 * max(1.0, abs(intr[j]))
            ip, im = intr.copy(), intr.copy()
            ip[j] += h
            im[j] -= h
            fd = (orc.ba_residual(model, pose, pt, ip, uv) - orc.ba_residual(model, pose, pt, im, uv)) / (2 * h)
            assert np.allclose(Ji[:, j], fd, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(Ji).max()))
        assert np.all(Ji[:, N_INTR[model]:] == 0)  # parameters a model does not use have zero columns


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_bundle_adjust_intrinsics_recovers_perturbed_intrinsics(orc, synth, model):
    # noise-free observations generated with the true intrinsics; start from intrinsics that are off by up to 2 % (and
    # perturbed poses / landmarks): the joint optimisation must drive the cost to ~0 and return to the true intrinsics
    # (the unused trailing parameters keep their values exactly)
    d = synth.ba_problem(80 + model, n_kf=5, n_lms=600, pix_noise=0.0, outlier_frac=0.0, integer_pixels=False)
    true = np.array([INTR8[model], INTR8[model]])
    uv = np.zeros_like(d["obs_uv"])
    gt_poses, gt_points = d["gt_poses"], d["gt_points"]
    for i in range(len(uv)):  # observations of the ground-truth scene through the TRUE model
        c, l = d["obs_cam"][i], d["obs_lm"][i]
        r = orc.ba_residual(model, gt_poses[c], gt_points[l], true[d["cam_intr"][c]], np.zeros(2))
        uv[i] = -r
    arr = orc.BaArrays(d["poses"], d["cam_fixed"], d["cam_intr"], true, d["points"], d["obs_cam"], d["obs_lm"], uv, (model, model))
    start = true.copy()
    start[:, :4] *= np.array([1.02, 0.985, 1.01, 0.99])
    start[:, 4:N_INTR[model]] *= 1.02
    arr.intr[...] = start
    s = orc.bundle_adjust_intrinsics(arr, max_iters=60)
    assert s.final_cost < 1e-10 * max(s.initial_cost, 1.0)
    assert np.array_equal(arr.intr[:, N_INTR[model]:], start[:, N_INTR[model]:])
