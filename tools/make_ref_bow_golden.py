#!/usr/bin/env python3
"""Writes tests/golden/dbow2_ref_streams.npz: operation streams for DBoW2::BowVector / FeatureVector and the
outputs of the REFERENCE's own classes on them (oracle/_ref/libdbow2_ref.so = the reference's BowVector.cpp and
FeatureVector.cpp compiled unmodified by `make -C oracle ref`).  Run in the build container, where /root/reference
exists; the fixture is data (inputs + expected outputs), so the pin also holds where the reference is absent.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402


def streams():
    """Deterministic operation streams: (name, ids, vals, ops, norm)."""
    out = []
    rng = np.random.default_rng(20261004)
    for case, (n, n_ids, norm) in enumerate([(0, 1, 1), (1, 1, 1), (1500, 400, 1), (1500, 1000000, 1), (4000, 37, 1),
                                              (1500, 400, 2), (1500, 400, 0), (257, 5, 1)]):
        ids = rng.integers(0, n_ids, n, dtype=np.uint32)
        # TF-IDF-like weights over many magnitudes, exact zeros and negative values included (fabs in normalize)
        vals = np.exp(rng.uniform(-12, 6, n)) * rng.choice([1.0, 1.0, 1.0, -1.0], n)
        vals[rng.random(n) < 0.02] = 0.0
        ops = (rng.random(n) < (0.5 if case % 2 else 0.0)).astype(np.uint8)
        out.append(("case%d" % case, ids, vals, ops, norm))
    return out


def fv_streams():
    rng = np.random.default_rng(7)
    out = []
    for case, (n, n_nodes) in enumerate([(0, 1), (1, 1), (1500, 90), (1500, 100000), (3000, 3)]):
        nodes = rng.integers(0, n_nodes, n, dtype=np.uint32)
        feats = rng.permutation(n).astype(np.uint32)
        out.append(("fv%d" % case, nodes, feats))
    return out


def main():
    orc = entry.load_oracle()
    assert orc.ref_lib() is not None, "oracle/_ref/libdbow2_ref.so missing and /root/reference absent"
    doc = {}
    for name, ids, vals, ops, norm in streams():
        oi, ov = orc.ref_bowvec_stream(ids, vals, ops, norm)
        doc.update({name + "_ids": ids, name + "_vals": vals, name + "_ops": ops, name + "_norm": np.int32(norm),
                    name + "_out_ids": oi, name + "_out_vals": ov})
    for name, nodes, feats in fv_streams():
        on, of = orc.ref_featvec_stream(nodes, feats)
        doc.update({name + "_nodes": nodes, name + "_feats": feats, name + "_out_nodes": on, name + "_out_feats": of})
    np.savez_compressed(ROOT / "tests" / "golden" / "dbow2_ref_streams.npz", **doc)
    print("wrote", len(doc), "arrays")


if __name__ == "__main__":
    main()
