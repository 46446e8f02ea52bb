"""CPU: the pose-graph restatement (oracle/orc_pgo.cpp; loop_closure_utils.h:446-587, reprojection.h:107-126).
[upstream] ceres::Solve / Sophus are absent: parity with their binaries is UNPINNED; pinned here are the
mathematical identities the restatement must satisfy -- log inverts exp, the dual-number Jacobians equal finite
differences along T * exp(delta), the normal equations equal J^T J, and the optimiser recovers a consistent graph."""
import numpy as np
import pytest


def _arr(orc, d, poses=None):
    return orc.PgoArrays(d["poses"] if poses is None else poses, d["node_fixed"], d["edge_a"], d["edge_b"], d["edge_meas"])


def test_log_inverts_exp(orc):
    rng = np.random.default_rng(0)
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    for scale in (1e-12, 1e-7, 1e-3, 0.5, 2.5, 3.1):
        for _ in range(20):
            xi = rng.normal(size=6)
            xi[3:] *= scale / np.linalg.norm(xi[3:])
            T = orc.se3_plus(ident, xi)          # exp(xi) (local_parameterization_se3.hpp:43-50)
            # (1 - cos theta) / theta^2 in exp loses digits for tiny angles, as in Sophus: 1e-10 absolute
            assert np.allclose(orc.se3_log(T), xi, rtol=1e-9, atol=1e-10)


def test_log_agrees_with_the_independent_numpy_formula(orc, synth):
    d = synth.pose_graph(1, 20, 5)
    for p in d["poses_gt"]:
        assert np.allclose(orc.se3_log(p), d["log"](p), rtol=1e-12, atol=1e-13)


def test_jacobians_equal_finite_differences(orc, synth):
    d = synth.pose_graph(2, 12, 4, meas_noise=0.05)
    rng = np.random.default_rng(3)
    for e in range(len(d["edge_a"])):
        pc, pn, m = d["poses"][d["edge_a"][e]], d["poses"][d["edge_b"][e]], d["edge_meas"][e]
        r, Jc, Jn = orc.pgo_residual_jacobian(pc, pn, m)
        assert np.allclose(r, d["log"](d["mul"](d["inv"](pc), pn)) - m, atol=1e-12)
        for k in range(6):
            dl = np.zeros(6)
            dl[k] = 1e-6
            rp = orc.pgo_residual_jacobian(orc.se3_plus(pc, dl), pn, m)[0]
            rm = orc.pgo_residual_jacobian(orc.se3_plus(pc, -dl), pn, m)[0]
            assert np.allclose((rp - rm) / 2e-6, Jc[:, k], atol=2e-8)
            rp = orc.pgo_residual_jacobian(pc, orc.se3_plus(pn, dl), m)[0]
            rm = orc.pgo_residual_jacobian(pc, orc.se3_plus(pn, -dl), m)[0]
            assert np.allclose((rp - rm) / 2e-6, Jn[:, k], atol=2e-8)
    assert rng is not None


def test_normal_equations_and_cost(orc, synth):
    d = synth.pose_graph(4, 15, 6, meas_noise=0.3, outlier_edges=2)
    a = _arr(orc, d)
    H, g, cost = orc.pgo_linearize(a, True, 1.0)
    n = 6 * a.n_free()
    assert H.shape == (n, n) and np.allclose(H, H.T, atol=1e-12)
    # rebuild from the per-edge blocks
    free = np.cumsum(d["node_fixed"] == 0) - 1
    free[d["node_fixed"] != 0] = -1
    H2, g2, c2 = np.zeros((n, n)), np.zeros(n), 0.0
    for e in range(len(d["edge_a"])):
        ia, ib = d["edge_a"][e], d["edge_b"][e]
        r, Jc, Jn = orc.pgo_residual_jacobian(d["poses"][ia], d["poses"][ib], d["edge_meas"][e])
        s = r @ r
        rho1 = 1.0 if s <= 1.0 else 1.0 / np.sqrt(s)
        c2 += 0.5 * (s if s <= 1.0 else 2 * np.sqrt(s) - 1.0)
        J = np.zeros((6, n))
        if free[ia] >= 0:
            J[:, 6 * free[ia]:6 * free[ia] + 6] = Jc
        if free[ib] >= 0:
            J[:, 6 * free[ib]:6 * free[ib] + 6] = Jn
        H2 += rho1 * J.T @ J
        g2 += rho1 * J.T @ r
    assert np.allclose(H, H2, rtol=1e-11, atol=1e-11) and np.allclose(g, g2, rtol=1e-11, atol=1e-11)
    assert abs(cost - c2) < 1e-11 * max(1.0, c2)


@pytest.mark.parametrize("seed,n_nodes", [(5, 30), (6, 120)])
def test_optimiser_recovers_a_consistent_graph(orc, synth, seed, n_nodes):
    d = synth.pose_graph(seed, n_nodes, n_nodes // 3, meas_noise=0.0, drift=0.03)
    a = _arr(orc, d)
    s = orc.pose_graph_optimize(a, True, 1.0, 50)
    assert s.final_cost < 1e-12 * max(1.0, s.initial_cost) or s.final_cost < 1e-14
    # every relative pose equals its measurement again (the gauge is free: compare relative poses)
    for e in range(len(d["edge_a"])):
        rel = d["log"](d["mul"](d["inv"](a.poses[d["edge_a"][e]]), a.poses[d["edge_b"][e]]))
        assert np.allclose(rel, d["edge_meas"][e], atol=1e-6)
    assert np.array_equal(a.poses[-1], d["poses"][-1])   # the fixed node did not move


def test_huber_keeps_outlier_edges_from_dominating(orc, synth):
    d = synth.pose_graph(7, 40, 15, meas_noise=0.001, drift=0.02, outlier_edges=3)
    a, b = _arr(orc, d), _arr(orc, d)
    s_h = orc.pose_graph_optimize(a, True, 0.02, 60)   # a width below the outliers' share of the error
    s_q = orc.pose_graph_optimize(b, False, 1.0, 60)
    assert s_h.final_cost < s_h.initial_cost and s_q.final_cost < s_q.initial_cost
    # with the robust loss the inlier edges are satisfied far better than with the quadratic loss
    def inlier_err(p):
        errs = []
        for e in range(len(d["edge_a"])):
            r = d["log"](d["mul"](d["inv"](p[d["edge_a"][e]]), p[d["edge_b"][e]])) - d["edge_meas"][e]
            errs.append(np.linalg.norm(r))
        return np.median(errs)
    assert inlier_err(a.poses) < 0.5 * inlier_err(b.poses)
