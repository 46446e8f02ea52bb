"""Developer probe: time of the ORB front end / compute_bow_vector on one 752x480 image (GPU vs the CPU restatement)."""
import importlib
import pathlib
import sys
import tempfile
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package()
orc = g.load_oracle()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = pkg.Context(0)
left, _ = synth.stereo_pair(7)
with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
    f.write(synth.vocabulary_text(5, 10, 4))
voc = pkg.Vocabulary(ctx, f.name)
for _ in range(3):
    ctx.orb_detect_describe(left, 1500)
    voc.compute_bow_vector(left, 1500, 4)
t = time.perf_counter()
for _ in range(20):
    kp, d = ctx.orb_detect_describe(left, 1500)
t_orb = (time.perf_counter() - t) / 20
t = time.perf_counter()
for _ in range(20):
    voc.compute_bow_vector(left, 1500, 4)
t_bow = (time.perf_counter() - t) / 20
t = time.perf_counter()
for _ in range(3):
    orc.orb_detect_describe(left, 1500)
t_cpu = (time.perf_counter() - t) / 3
print("ORB front end: %.3f ms GPU (%d keypoints), compute_bow_vector %.3f ms, CPU restatement %.1f ms" % (1e3 * t_orb, len(kp), 1e3 * t_bow, 1e3 * t_cpu))
