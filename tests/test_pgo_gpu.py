"""GPU parity: pose graph optimisation (visual-slam_amd/csrc/pgo.hip) vs oracle/orc_pgo.cpp -- the numerical core of
pose_graph_optimization (loop_closure_utils.h:446-587).  Floating point: the normal equations agree to 1e-9
relative (fp64 atomics change the summation order), the LM trajectories coincide (same iteration counts and
termination), final poses to 1e-7, costs to 1e-6 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _arr(orc, d):
    return orc.PgoArrays(d["poses"], d["node_fixed"], d["edge_a"], d["edge_b"], d["edge_meas"])


@pytest.mark.parametrize("huber", [True, False])
def test_normal_equations_match_the_oracle(ctx, orc, synth, huber):
    d = synth.pose_graph(11, 25, 10, meas_noise=0.2, outlier_edges=2)
    a = _arr(orc, d)
    H, g, cost = ctx.pgo_linearize(a, huber, 0.5)
    oH, og, ocost = orc.pgo_linearize(a, huber, 0.5)
    scale = np.abs(oH).max()
    assert np.abs(H - oH).max() < 1e-9 * scale and np.abs(g - og).max() < 1e-9 * max(1.0, np.abs(og).max())
    assert abs(cost - ocost) < 1e-12 * max(1.0, ocost)


@pytest.mark.parametrize("seed,n_nodes,noise,outliers", [(21, 30, 0.0, 0), (22, 60, 0.01, 0), (23, 150, 0.003, 4), (24, 400, 0.002, 0)])
def test_pose_graph_optimize_matches_the_oracle(ctx, orc, synth, seed, n_nodes, noise, outliers):
    d = synth.pose_graph(seed, n_nodes, n_nodes // 3, meas_noise=noise, drift=0.02, outlier_edges=outliers)
    a, b = _arr(orc, d), _arr(orc, d)
    s = ctx.pose_graph_optimize(a, True, 1.0, 20)
    os_ = orc.pose_graph_optimize(b, True, 1.0, 20)
    assert s.iterations == os_.iterations and s.termination == os_.termination and s.successful_steps == os_.successful_steps
    assert abs(s.initial_cost - os_.initial_cost) <= 1e-9 * max(1.0, os_.initial_cost)
    assert abs(s.final_cost - os_.final_cost) <= 1e-6 * max(os_.final_cost, 1e-12) + 1e-15
    assert np.abs(a.poses - b.poses).max() < 1e-7
    assert s.final_cost < s.initial_cost
    assert np.array_equal(a.poses[-1], d["poses"][-1])   # the fixed node did not move


@pytest.mark.parametrize("seed,n_nodes,window,noise", [(31, 400, 6, 0.002), (32, 500, 1, 0.003), (33, 120, 3, 0.002), (34, 300, 12, 0.001)])
def test_banded_pose_graphs_take_the_band_solvers_and_match_the_oracle(ctx, orc, synth, seed, n_nodes, window, noise):
    # graphs with odometry + covisibility edges inside a time window and the loop edge (no random long-range edges): the
    # normal equations are kept in cyclic band storage and solved by the ring form of the block cyclic reduction (400 /
    # 500 / 300 nodes), or in linear band storage when the ring has too few blocks (120 nodes) -- same LM trajectory as
    # the oracle's dense solve, and as the dense device path
    d = synth.pose_graph(seed, n_nodes, 0, meas_noise=noise, drift=0.02, window=window)
    a, b, c = _arr(orc, d), _arr(orc, d), _arr(orc, d)
    s = ctx.pose_graph_optimize(a, True, 1.0, 20)
    os_ = orc.pose_graph_optimize(b, True, 1.0, 20)
    assert s.iterations == os_.iterations and s.termination == os_.termination and s.successful_steps == os_.successful_steps
    assert abs(s.initial_cost - os_.initial_cost) <= 1e-9 * max(1.0, os_.initial_cost)
    assert abs(s.final_cost - os_.final_cost) <= 1e-6 * max(os_.final_cost, 1e-12) + 1e-15
    assert np.abs(a.poses - b.poses).max() < 1e-7
    assert s.final_cost < s.initial_cost
    ctx.set_diagnostic("ba_force_dense", 1)
    try:
        sd = ctx.pose_graph_optimize(c, True, 1.0, 20)
    finally:
        ctx.set_diagnostic("ba_force_dense", 0)
    assert (sd.iterations, sd.termination, sd.successful_steps) == (s.iterations, s.termination, s.successful_steps)
    assert np.abs(a.poses - c.poses).max() < 1e-8


def test_consistent_graph_converges_to_zero_cost(ctx, orc, synth):
    d = synth.pose_graph(31, 80, 30, meas_noise=0.0, drift=0.03)
    a = _arr(orc, d)
    s = ctx.pose_graph_optimize(a, True, 1.0, 50)
    assert s.final_cost < 1e-14
    for e in range(len(d["edge_a"])):
        rel = d["log"](d["mul"](d["inv"](a.poses[d["edge_a"][e]]), a.poses[d["edge_b"][e]]))
        assert np.allclose(rel, d["edge_meas"][e], atol=1e-6)


def test_degenerate_inputs(ctx, orc, vsl, synth):
    d = synth.pose_graph(41, 10, 3)
    a = _arr(orc, d)
    a.node_fixed[:] = 1                       # nothing to optimise
    before = a.poses.copy()
    s = ctx.pose_graph_optimize(a, True, 1.0, 20)
    assert s.iterations == 0 and np.array_equal(a.poses, before)
    bad = _arr(orc, d)
    bad.edge_a[0] = 99
    with pytest.raises(vsl.VslError):
        ctx.pose_graph_optimize(bad, True, 1.0, 20)
    loop = _arr(orc, d)
    loop.edge_b[1] = loop.edge_a[1]
    with pytest.raises(vsl.VslError):
        ctx.pose_graph_optimize(loop, True, 1.0, 20)
