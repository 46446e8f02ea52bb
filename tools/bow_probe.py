"""K8 / K9 at the reference's vocabulary shape (k = 10, L = 6: 1,111,111 nodes): device time of `transform` for the
1500 ORB descriptors of one image and of `score_batch` for M candidates, next to the oracle on one core.
    python tools/bow_probe.py [--M 100,1000,10000]"""
import argparse
import importlib
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", default="100,1000,10000")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    vsl = entry.load_package()
    synth = importlib.import_module("visual_slam_amd.synth")
    path = "/tmp/vsl_voc_k10L6_s7.txt"
    t = time.perf_counter()
    if not os.path.exists(path):
        synth.write_vocabulary_text(path, 10, 6, *synth.vocabulary_arrays(7, 10, 6))
    print("vocabulary text: %.1f s, %.1f MB" % (time.perf_counter() - t, os.path.getsize(path) / 1e6), flush=True)
    ctx = vsl.Context(0)
    t = time.perf_counter()
    voc = ctx.load_vocabulary(path)
    print("gpu load %.2f s" % (time.perf_counter() - t), voc.info(), flush=True)
    pool = []
    descs = []
    for s in range(16):
        left, _ = synth.stereo_pair(100 + s)
        d = ctx.orb_detect_describe(left, 1500)[-1]
        descs.append(d)
        pool.append(voc.transform(d, 4)[:2])
    print("descriptors per image", [len(d) for d in descs[:4]], "nnz", [len(p[0]) for p in pool[:4]], flush=True)
    ctx.set_profiling(True)
    for rep in range(3):
        ctx.reset_profiling()
        for _ in range(20):
            voc.transform(descs[0], 4)
        st = ctx.stage_ms()
        print("transform: stage ms per call", {k: round(v[0] / v[1], 5) for k, v in st.items() if v[1]}, flush=True)
    t = time.perf_counter()
    for _ in range(20):
        voc.transform(descs[0], 4)
    print("transform host wall per call %.3f ms" % (1e3 * (time.perf_counter() - t) / 20), flush=True)
    q = pool[0]
    for M in [int(x) for x in args.M.split(",")]:
        cands = [pool[1 + (i % 15)] for i in range(M)]
        ctx.bow_score_batch(q[0], q[1], cands)
        ctx.reset_profiling()
        t = time.perf_counter()
        for _ in range(5):
            sc = ctx.bow_score_batch(q[0], q[1], cands)
        wall = 1e3 * (time.perf_counter() - t) / 5
        st = ctx.stage_ms()
        tot = sum(len(c[0]) for c in cands)
        ms = st["bow_score"][0] / st["bow_score"][1]
        print("score M=%d: kernel %.4f ms, wall (host arrays -> upload -> kernel -> scores) %.3f ms, bytes %d -> %.1f GB/s"
              % (M, ms, wall, 12 * (len(q[0]) + tot) + 8 * M, (12 * (len(q[0]) + tot) + 8 * M) / ms / 1e6), flush=True)
    if not args.no_oracle:
        orc = entry.load_oracle()
        ov = orc.Vocabulary(path)
        t = time.perf_counter()
        for _ in range(5):
            o = ov.transform(descs[0], 4)
        print("oracle transform %.3f ms" % (1e3 * (time.perf_counter() - t) / 5), flush=True)
        g = voc.transform(descs[0], 4)
        print("transform equal:", all(np.array_equal(a, b) for a, b in zip(o, g)), flush=True)
        t = time.perf_counter()
        osc = [orc.bow_score_l1(q[0], q[1], c[0], c[1]) for c in pool[1:]]
        print("oracle score %.4f ms per candidate; equal: %s" % (1e3 * (time.perf_counter() - t) / 15,
                                                                   np.array_equal(np.array(osc), ctx.bow_score_batch(q[0], q[1], pool[1:]))), flush=True)


if __name__ == "__main__":
    main()
