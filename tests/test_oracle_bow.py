"""CPU: known-answer tests pinning the oracle's DBoW2 restatement (TemplatedVocabulary.h transform,
BowVector.cpp, FeatureVector.cpp, ScoringObject.cpp L1)."""
import numpy as np
import pytest


def _write_voc(tmp_path, text, name="voc.txt"):
    p = tmp_path / name
    p.write_text(text)
    return p


def _tiny_voc_text():
    # k=2, L=2: root -> A(id1: all zeros), B(id2: all ones); leaves under A: a0 (zeros, w=1.0),
    # a1 (first byte 0xFF, w=2.0); under B: b0 (ones, w=0 => stopped word), b1 (ones except last byte, w=4.0)
    z = " ".join(["0"] * 32)
    o = " ".join(["255"] * 32)
    a1 = " ".join(["255"] + ["0"] * 31)
    b1 = " ".join(["255"] * 31 + ["0"])
    lines = ["2 2 0 0", "0 0 %s 0" % z, "0 0 %s 0" % o, "1 1 %s 1.0" % z, "1 1 %s 2.0" % a1,
             "2 1 %s 0.0" % o, "2 1 %s 4.0" % b1]
    return "\n".join(lines) + "\n"


def test_l1_score_identities(orc):
    ids = np.array([3, 10, 42], np.uint32)
    v = np.array([0.5, 0.25, 0.25])
    assert orc.bow_score_l1(ids, v, ids, v) == 1.0                     # identical -> 1
    assert orc.bow_score_l1(ids, v, ids + 100, v) == 0.0               # disjoint -> 0
    # one shared word: -( |.5-.25| - .5 - .25 ) / 2 = 0.25
    assert orc.bow_score_l1(ids, v, np.array([3], np.uint32), np.array([0.25])) == 0.25
    assert orc.bow_score_l1(ids[:0], v[:0], ids, v) == 0.0


def test_transform_tiny_tree(orc, tmp_path):
    voc = orc.Vocabulary(_write_voc(tmp_path, _tiny_voc_text()))
    assert voc.info() == (2, 2, 7, 4)
    f = np.zeros((5, 32), np.uint8)
    f[1, 0] = 0xFF            # -> a1 (word 1, weight 2)
    f[2, :] = 0xFF            # -> b0 (word 2, weight 0: stopped, dropped)
    f[3, :31] = 0xFF          # -> b1 (word 3, weight 4)
    f[4, 0] = 0x0F            # 4 bits from a0, 4 bits from a1: tie -> FIRST child (a0) wins (strict <)
    ids, vals, fn, ff = voc.transform(f, levelsup=1)
    # words: 0 (features 0 and 4: 1+1=2), 1 (feature 1: 2), 3 (feature 3: 4); L1 norm 8
    assert ids.tolist() == [0, 1, 3]
    assert vals.tolist() == [2 / 8, 2 / 8, 4 / 8]
    # feature vector at level L - levelsup = 1: node 1 (A) holds features 0,1,4; node 2 (B) holds 3
    assert list(zip(fn.tolist(), ff.tolist())) == [(1, 0), (1, 1), (1, 4), (2, 3)]
    # levelsup >= L: everything under the root (node 0)
    _, _, fn0, ff0 = voc.transform(f, levelsup=4)
    assert set(fn0.tolist()) == {0} and ff0.tolist() == [0, 1, 3, 4]


def test_transform_random_tree_properties(orc, synth, tmp_path):
    voc = orc.Vocabulary(_write_voc(tmp_path, synth.vocabulary_text(1, k=10, L=3)))
    k, L, n_nodes, n_words = voc.info()
    assert (k, L, n_nodes, n_words) == (10, 3, 1 + 10 + 100 + 1000, 1000)
    rng = np.random.default_rng(0)
    f = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    ids, vals, fn, ff = voc.transform(f, levelsup=1)
    assert np.all(np.diff(ids.astype(np.int64)) > 0)
    assert vals.sum() == pytest.approx(1.0, abs=1e-12) and np.all(vals > 0)
    assert len(ff) <= 700 and len(np.unique(ff)) == len(ff)
    # self-score 1, score symmetric within rounding, in [0, 1]
    ids2, vals2, _, _ = voc.transform(f[::2], levelsup=1)
    s = orc.bow_score_l1(ids, vals, ids2, vals2)
    assert 0.0 < s < 1.0 and orc.bow_score_l1(ids, vals, ids, vals) == pytest.approx(1.0, abs=1e-12)
    assert s == pytest.approx(orc.bow_score_l1(ids2, vals2, ids, vals), abs=1e-12)


def test_blank_lines_are_skipped(orc, tmp_path):
    voc = orc.Vocabulary(_write_voc(tmp_path, _tiny_voc_text() + "\n\n"))
    assert voc.info() == (2, 2, 7, 4)
    with pytest.raises(IOError):
        orc.Vocabulary(_write_voc(tmp_path, "99 2 0 0\n", "bad.txt"))
