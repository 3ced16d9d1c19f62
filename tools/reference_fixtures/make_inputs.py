#!/usr/bin/env python3
"""Writes the inputs tools/reference_fixtures/dump_reference_fixtures.rs reads, from the golden
fixtures of this repository: every PCM clip of tests/golden/features.npz and the 'mid' network
case of tests/golden/net.npz (parameters in the reference's model.npz layout, lib.rs:1081-1130).

    python tools/reference_fixtures/make_inputs.py <output dir>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out):
    os.makedirs(out, exist_ok=True)
    feats = np.load(os.path.join(ROOT, "tests", "golden", "features.npz"))
    clips = sorted(k[4:] for k in feats.files if k.startswith("pcm_"))
    for c in clips:
        np.save(os.path.join(out, f"pcm_{c}.npy"), feats[f"pcm_{c}"].astype(np.int16))
    with open(os.path.join(out, "clips.txt"), "w") as f:
        f.write("\n".join(clips) + "\n")
    net = np.load(os.path.join(ROOT, "tests", "golden", "net.npz"))
    i, h1, h2, c = [int(v) for v in net["mid_dims"]]
    w1, b1, w2, b2, w3, b3 = [net[f"mid_p{k}"] for k in range(6)]
    members = {"w1": w1.reshape(i, h1), "b1": b1.reshape(h1), "w2": w2.reshape(h1, h2), "b2": b2.reshape(h2),
               "sample_rate": np.array([44100], np.int64), "bits": np.array([16], np.int64),
               "num_speakers": np.array([c], np.int64)}
    w3 = w3.reshape(h2, c)
    for k in range(c):  # one column and one bias per class, 1-based names (lib.rs:1099-1107)
        members[f"w3_{k + 1}"] = np.ascontiguousarray(w3[:, k])
        members[f"b3_{k + 1}"] = np.array([b3.reshape(c)[k]], np.float32)
        members[f"speaker_{k}_files"] = np.zeros(0, np.uint8)
    np.savez(os.path.join(out, "model_in.npz"), **members)
    np.save(os.path.join(out, "x.npy"), net["mid_x"].astype(np.float32).reshape(-1, i))
    np.save(os.path.join(out, "target.npy"), net["mid_target"].astype(np.float32))
    np.save(os.path.join(out, "lr.npy"), np.array([0.05], np.float32))
    print(f"wrote {len(clips)} clips and the 'mid' network case to {out}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "reference_fixture_inputs")
