#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- run ONCE in the build container (needs /root/reference for the
EuRoC-shaped input images; nothing under tests/ reads the reference at run time).

Inputs : 16 stereo pairs of the reference's data/euroc_V1 (752x480 8-bit gray JPEG; the first two of the directory,
         four spread over the 82 listed in its timestamps.txt, and the run of ten consecutive 20 Hz frames at its end
         -- the only consecutive frames the reference ships, also the real-image sequence of the headless harness
         test), decoded here
         with PIL and stored as raw pixels (JPEG decoders differ between versions, so the pixels --
         not the JPEGs -- are the fixture).
Outputs: what the CPU oracle (oracle/, a restatement of include/visnav/keypoints.h) produces for
         detectKeypointsAndDescriptors(img, 1500, true) on each image and matchDescriptors(L, R, 70,
         1.2) on each pair.  The reference binary itself cannot be built offline, so
         these vectors pin the oracle against accidental change and pin the HIP path against the
         oracle; parity with real OpenCV's goodFeaturesToTrack stays "unpinned" (DESIGN.md).
"""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

def pick_pairs(ref):
    stamps = sorted({f.name.split("_")[0] for f in (ref / "data/euroc_V1").glob("*_0.jpg")})
    assert len(stamps) == 100
    first = ["1403715273262142976", "1403715308112143104"]          # pair0 / pair1 of round 1 (kept)
    spread = [stamps[i] for i in (10, 30, 50, 70)]
    tail = stamps[-12:-2]                                            # ten frames 50 ms apart
    out = first + [s for s in spread + tail if s not in first]
    assert len(out) == 16
    return out


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    orc = entry.load_oracle()
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    for k, stamp in enumerate(pick_pairs(ref)):
        imgs = [np.array(Image.open(ref / "data/euroc_V1" / ("%s_%d.jpg" % (stamp, c)))) for c in (0, 1)]
        rec = {"stamp": np.array(stamp)}
        descs = []
        for c, img in enumerate(imgs):
            assert img.shape == (480, 752) and img.dtype == np.uint8
            xy, ang, desc = orc.detect_describe(img, 1500, True)
            m01, m10 = orc.patch_moments(img, xy)
            resp = orc.min_eig_response(img)
            rec["img%d" % c] = img
            rec["xy%d" % c] = xy.astype(np.int32)
            rec["angle_bits%d" % c] = ang.view(np.uint64)
            rec["m01_%d" % c] = m01
            rec["m10_%d" % c] = m10
            rec["desc%d" % c] = desc
            rec["resp_max_bits%d" % c] = np.array(resp.max(), np.float32).view(np.uint32)
            rec["resp_bits_xor%d" % c] = np.bitwise_xor.reduce(resp.view(np.uint32).ravel())
            rec["resp_bits_sum%d" % c] = resp.view(np.uint32).astype(np.uint64).sum()
            descs.append(desc)
        rec["matches"] = orc.match_descriptors(descs[0], descs[1], 70, 1.2)
        np.savez_compressed(out_dir / ("euroc_pair%d.npz" % k), **rec)
        print("pair", k, stamp, "kp", len(rec["xy0"]), len(rec["xy1"]), "matches", len(rec["matches"]))


if __name__ == "__main__":
    main()
