"""GPU parity: K8 vocabulary transform and K9 batched L1 score (visual-slam_amd/csrc/bow.hip) vs the
oracle's DBoW2 restatement.  Integer tree descent + doubles summed in the reference's order:
bit-exact (ids identical, values compared as 64-bit patterns)."""
import numpy as np
import pytest

from test_oracle_bow import _tiny_voc_text, _write_voc

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def vocs(ctx, orc, synth, tmp_path_factory):
    d = tmp_path_factory.mktemp("voc")
    out = {}
    for name, text in (("tiny", _tiny_voc_text()), ("k10L3", synth.vocabulary_text(2, k=10, L=3)),
                       ("k10L4", synth.vocabulary_text(3, k=10, L=4)), ("k20L2", synth.vocabulary_text(4, k=20, L=2)),
                       ("k3L6", synth.vocabulary_text(5, k=3, L=6))):
        p = _write_voc(d, text, name + ".txt")
        out[name] = (ctx.load_vocabulary(p), orc.Vocabulary(p))
    return out


@pytest.mark.parametrize("name", ["tiny", "k10L3", "k10L4", "k20L2", "k3L6"])
@pytest.mark.parametrize("n,levelsup", [(1, 1), (63, 2), (1500, 4), (2049, 0)])
def test_transform_bit_exact(vocs, name, n, levelsup):
    gv, ov = vocs[name]
    assert gv.info() == ov.info()
    rng = np.random.default_rng(n + levelsup)
    f = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    f[: n // 3] = f[n // 3: 2 * (n // 3)]  # repeated descriptors -> words hit several times
    g = gv.transform(f, levelsup)
    e = ov.transform(f, levelsup)
    assert np.array_equal(g[0], e[0])
    assert np.array_equal(_bits(g[1]), _bits(e[1]))
    assert np.array_equal(g[2], e[2]) and np.array_equal(g[3], e[3])


def test_transform_of_real_descriptors(ctx, orc, vsl, vocs, synth):
    # the reference feeds cv::ORB descriptors; here rBRIEF descriptors converted with converter.h's rule
    left, _ = synth.stereo_pair(5)
    _, _, desc = orc.detect_describe(left, 1500, True)
    import ctypes as C
    b = np.zeros((len(desc), 32), np.uint8)
    ctx.L.vsl_desc_bitset_to_bytes(desc.ctypes.data_as(vsl.u64p), len(desc), b.ctypes.data_as(vsl.u8p))
    assert np.array_equal(b, orc.bitset_to_bytes(desc))
    back = np.zeros_like(desc)
    ctx.L.vsl_desc_bytes_to_bitset(b.ctypes.data_as(vsl.u8p), len(desc), back.ctypes.data_as(vsl.u64p))
    assert np.array_equal(back, desc)  # the round trip the reference's own test asserts
    gv, ov = vocs["k10L4"]
    g, e = gv.transform(b, 4), ov.transform(b, 4)
    assert np.array_equal(g[0], e[0]) and np.array_equal(_bits(g[1]), _bits(e[1]))
    assert np.array_equal(g[2], e[2]) and np.array_equal(g[3], e[3])


def test_score_batch_bit_exact(ctx, orc, vocs):
    gv, ov = vocs["k10L3"]
    rng = np.random.default_rng(9)
    q = ov.transform(rng.integers(0, 256, (1500, 32), dtype=np.uint8), 4)
    cands = []
    for m in range(40):
        n = int(rng.integers(1, 1500))
        c = ov.transform(rng.integers(0, 256, (n, 32), dtype=np.uint8), 4)
        cands.append((c[0], c[1]))
    cands.append((q[0], q[1]))                                        # identical -> 1
    cands.append((np.zeros(0, np.uint32), np.zeros(0)))                # empty -> 0
    cands.append((q[0] + np.uint32(10 ** 6), q[1]))                    # disjoint -> 0
    got = ctx.bow_score_batch(q[0], q[1], cands)
    exp = np.array([orc.bow_score_l1(q[0], q[1], c[0], c[1]) for c in cands])
    assert np.array_equal(_bits(got), _bits(exp))
    assert got[-3] == pytest.approx(1.0, abs=1e-12) and got[-2] == 0.0 and got[-1] == 0.0
    assert len(ctx.bow_score_batch(q[0], q[1], [])) == 0


def test_voc_errors(ctx, vsl, tmp_path):
    with pytest.raises(vsl.VslError) as e:
        ctx.load_vocabulary(tmp_path / "missing.txt")
    assert e.value.code == -6
    p = _write_voc(tmp_path, "10 3 1 0\n", "l2.txt")  # L2 scoring: not implemented
    with pytest.raises(vsl.VslError) as e:
        ctx.load_vocabulary(p)
    assert e.value.code == -1
