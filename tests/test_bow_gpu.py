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


def test_transform_with_64_bit_sort_keys(ctx, vocs):
    # the assembly kernel sorts 32-bit (id, feature) keys where they fit; the 64-bit variants must agree (both capacities)
    gv, ov = vocs["k10L4"]
    rng = np.random.default_rng(11)
    for n in (1500, 5000):
        f = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        f[: n // 4] = f[n // 4: 2 * (n // 4)]
        e = ov.transform(f, 2)
        ctx.set_diagnostic("bow_keys64", 1)
        try:
            g = gv.transform(f, 2)
        finally:
            ctx.set_diagnostic("bow_keys64", 0)
        assert np.array_equal(g[0], e[0]) and np.array_equal(_bits(g[1]), _bits(e[1]))
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


@pytest.fixture(scope="module")
def orbvoc_shape(ctx, orc, synth, tmp_path_factory):
    """The reference's vocabulary SHAPE: k = 10, L = 6 -> 1,111,111 nodes, 1,000,000 words, 146 MB of text
    (TemplatedVocabulary.h:1338-1424; ORBvoc.txt itself is a missing blob, the tree here is synthetic)."""
    p = tmp_path_factory.mktemp("orbvoc") / "k10L6.txt"
    synth.write_vocabulary_text(p, 10, 6, *synth.vocabulary_arrays(7, 10, 6))
    return ctx.load_vocabulary(p), orc.Vocabulary(p)


def test_transform_bit_exact_on_the_orb_vocabulary_shape(ctx, orc, synth, orbvoc_shape):
    # the largest tree any test loaded had 11 k nodes.  Random descriptors, ORB descriptors of a frame
    # (what compute_bow_vector feeds, keypoints.h:250-253), repeated descriptors, both capacity variants of the kernel
    gv, ov = orbvoc_shape
    assert gv.info() == ov.info() == (10, 6, 1111111, 1000000)
    rng = np.random.default_rng(3)
    left, _ = synth.stereo_pair(21)
    orb = ctx.orb_detect_describe(left, 1500)[-1]
    rnd = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    rnd[1000:2000] = rnd[:1000]
    for f, levelsup in ((orb, 4), (orb[:1], 4), (rnd, 4), (rnd[:777], 2), (orb, 0), (orb, 6), (orb, 9)):
        g, e = gv.transform(f, levelsup), ov.transform(f, levelsup)
        assert np.array_equal(g[0], e[0]) and np.array_equal(_bits(g[1]), _bits(e[1]))
        assert np.array_equal(g[2], e[2]) and np.array_equal(g[3], e[3])
        assert len(g[0]) > 0.9 * len(np.unique(f, axis=0)) * 0.95   # a million words: nearly every feature its own word


def test_bow_database_scores_equal_the_pairwise_scores(ctx, vsl, orc, synth, orbvoc_shape):
    # vsl_bowdb_*: vectors appended once, scored by index -- bit-equal to ScoringObject.cpp:23-68 pair by pair,
    # in any candidate order, with repeats, after the store has grown past its initial capacity
    gv, ov = orbvoc_shape
    vecs = []
    for s in range(12):
        left, right = synth.stereo_pair(40 + s)
        for img in (left, right):
            d = ctx.orb_detect_describe(img, 1500)[-1]
            vecs.append(gv.transform(d, 4)[:2])
    vecs.append((np.zeros(0, np.uint32), np.zeros(0)))   # an empty vector
    db = vsl.BowDatabase(ctx, cap_entries=4096, cap_vectors=4)   # forces both growth paths
    for i, (ids, vals) in enumerate(vecs):
        assert db.append(ids, vals) == i
    assert db.info() == (len(vecs), sum(len(v[0]) for v in vecs))
    rng = np.random.default_rng(0)
    for qi in (0, 1, 7):
        q = vecs[qi]
        exp_all = np.array([orc.bow_score_l1(q[0], q[1], c[0], c[1]) for c in vecs])
        assert np.array_equal(_bits(db.score(q[0], q[1])), _bits(exp_all))
        idx = rng.integers(0, len(vecs), 300).astype(np.int32)
        assert np.array_equal(_bits(db.score(q[0], q[1], idx)), _bits(exp_all[idx]))
        assert np.array_equal(_bits(ctx.bow_score_batch(q[0], q[1], vecs)), _bits(exp_all))
    assert exp_all[7] == pytest.approx(1.0, abs=1e-12) and exp_all[-1] == 0.0
    q = vecs[0]
    with pytest.raises(vsl.VslError):
        db.score(q[0], q[1], np.array([len(vecs)], np.int32))
    with pytest.raises(vsl.VslError):
        db.append(np.array([5, 5], np.uint32), np.array([0.5, 0.5]))
    db.close()


def test_score_kernel_forms_agree_bit_for_bit(ctx, orc):
    # round 4: <= 256 candidates of <= 4096 words against a query of <= 4096 words take the workgroup-per-candidate
    # kernel (one round of searches by 1024 threads, ordered sum by one wavefront); anything else the wave-per-candidate
    # kernel.  Both against the oracle, bit for bit: candidate lengths around the staging limits (1, 1023, 1024, 1025,
    # 4096, 4097 -- the last one sends the whole batch to the other kernel), heavy and empty overlaps, the diagnostic switch.
    rng = np.random.default_rng(14)
    q_ids = np.sort(rng.choice(10 ** 6, 3000, replace=False)).astype(np.uint32)
    q_vals = rng.random(3000)
    q_vals /= q_vals.sum()

    def cand(n, share):
        ids = rng.choice(10 ** 6, n, replace=False).astype(np.uint32)
        k = min(int(share * n), len(q_ids))
        ids[:k] = rng.choice(q_ids, k, replace=False)
        ids = np.unique(ids)
        v = rng.random(len(ids))
        return ids, v / v.sum()

    base = [cand(n, sh) for n, sh in ((1, 1.0), (1023, 0.5), (1024, 0.0), (1025, 0.9), (2500, 0.3), (4000, 0.7))]
    base.append((np.zeros(0, np.uint32), np.zeros(0)))
    for extra in ([], [cand(4300, 0.5)]):          # with the long candidate no batch member may use the workgroup form
        cands = base + extra
        exp = np.array([orc.bow_score_l1(q_ids, q_vals, c[0], c[1]) for c in cands])
        got = ctx.bow_score_batch(q_ids, q_vals, cands)
        assert np.array_equal(_bits(got), _bits(exp))
        ctx.set_diagnostic("bow_no_wg_score", 1)
        try:
            got2 = ctx.bow_score_batch(q_ids, q_vals, cands)
        finally:
            ctx.set_diagnostic("bow_no_wg_score", 0)
        assert np.array_equal(_bits(got2), _bits(exp))
    # a query beyond the workgroup form's 4096 words
    q2 = np.sort(rng.choice(10 ** 6, 5000, replace=False)).astype(np.uint32)
    v2 = rng.random(5000)
    v2 /= v2.sum()
    got = ctx.bow_score_batch(q2, v2, base)
    assert np.array_equal(_bits(got), _bits(np.array([orc.bow_score_l1(q2, v2, c[0], c[1]) for c in base])))


def test_score_with_a_query_too_large_for_lds(ctx, orc):
    # > 8192 query words: the global-memory kernel; and exactly at the LDS kernel's limit
    rng = np.random.default_rng(4)
    for q_n in (8192, 8193, 20000):
        q_ids = np.sort(rng.choice(10 ** 6, q_n, replace=False)).astype(np.uint32)
        q_vals = rng.random(q_n)
        q_vals /= q_vals.sum()
        cands = []
        for n in (1, 100, 5000):
            ids = np.sort(rng.choice(10 ** 6, n, replace=False)).astype(np.uint32)
            ids[: n // 2] = np.sort(rng.choice(q_ids, n // 2, replace=False))[: n // 2]
            ids = np.unique(ids)
            v = rng.random(len(ids))
            cands.append((ids, v / v.sum()))
        got = ctx.bow_score_batch(q_ids, q_vals, cands)
        exp = np.array([orc.bow_score_l1(q_ids, q_vals, c[0], c[1]) for c in cands])
        assert np.array_equal(_bits(got), _bits(exp))


def test_wide_child_lists_use_the_wide_descent_groups(ctx, orc, tmp_path):
    # a node may list more children than the header's k (the reference's loader does not check): 40 children under the
    # root -> 64 lanes per descriptor; 20 -> 32 lanes (k20L2 above)
    rng = np.random.default_rng(6)
    lines = ["20 2 0 0"]
    for _ in range(40):
        lines.append("0 0 %s 0" % " ".join(str(int(v)) for v in rng.integers(0, 256, 32)))
    for parent in range(1, 41):
        for _ in range(3):
            lines.append("%d 1 %s %.3f" % (parent, " ".join(str(int(v)) for v in rng.integers(0, 256, 32)), rng.uniform(0.5, 4)))
    p = _write_voc(tmp_path, "\n".join(lines) + "\n", "wide.txt")
    gv, ov = ctx.load_vocabulary(p), orc.Vocabulary(p)
    f = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    for levelsup in (0, 1, 2):
        g, e = gv.transform(f, levelsup), ov.transform(f, levelsup)
        assert np.array_equal(g[0], e[0]) and np.array_equal(_bits(g[1]), _bits(e[1]))
        assert np.array_equal(g[2], e[2]) and np.array_equal(g[3], e[3])


def test_voc_text_parser_matches_the_stream_parser(ctx, orc, tmp_path):
    # the hand-written parser of vsl_voc_load_text against the oracle's stringstream reader on awkward text: CRLF line
    # ends, tabs, several blanks, a truncated line (missing bytes / weight -> zeros), blank lines, no final newline
    z = " ".join(["0"] * 32)
    o = " ".join(["255"] * 32)
    text = ("2 2 0 0\r\n0 0 %s 0\r\n0  0\t%s   0\n\n1 1 %s 1.5\n1 1 255 255\n2 1 %s 2.25e0\n   \n2 1 %s 4"
            % (z, o, z, o, " ".join(["255"] * 31 + ["7"])))
    p = _write_voc(tmp_path, text, "awkward.txt")
    gv, ov = ctx.load_vocabulary(p), orc.Vocabulary(p)
    assert gv.info() == ov.info() == (2, 2, 7, 4)
    rng = np.random.default_rng(2)
    f = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    f[:50] = 0
    f[50:100] = 255
    g, e = gv.transform(f, 1), ov.transform(f, 1)
    assert np.array_equal(g[0], e[0]) and np.array_equal(_bits(g[1]), _bits(e[1]))
    assert np.array_equal(g[2], e[2]) and np.array_equal(g[3], e[3])


def test_voc_errors(ctx, vsl, tmp_path):
    with pytest.raises(vsl.VslError) as e:
        ctx.load_vocabulary(tmp_path / "missing.txt")
    assert e.value.code == -6
    p = _write_voc(tmp_path, "10 3 1 0\n", "l2.txt")  # L2 scoring: not implemented
    with pytest.raises(vsl.VslError) as e:
        ctx.load_vocabulary(p)
    assert e.value.code == -1
