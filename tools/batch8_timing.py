"""Wall and kernel time of the reference's batch-of-8 training schedule (pretrain_from_features,
lib.rs:582-628) through the resident step kernel.  Usage: batch8_timing.py [epochs ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import streamz_amd as sz

rng = np.random.default_rng(0)
wins = rng.standard_normal((6614, 60)).astype(np.float32)
net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=3)
sz.pretrain_from_features(net, wins[:64], 1, 16, 1, 0.01, 0.2, 8, seed=1)
for ep in [int(v) for v in sys.argv[1:]] or [1, 10]:
    sz.profile_enable(True)
    sz.profile_report()
    t0 = time.perf_counter()
    loss = sz.pretrain_from_features(net, wins, 1, 16, ep, 0.01, 0.2, 8, seed=2)
    dt = time.perf_counter() - t0
    rep = sz.profile_report()
    sz.profile_enable(False)
    steps = ep * 827
    print(f"epochs {ep}: wall {dt*1e3:.1f} ms, kernel {rep['steps'][1]:.1f} ms, {rep['steps'][1]*1e3/steps:.2f} us/step, "
          f"{ep*6614/dt:.0f} windows/s wall, loss {loss:.4f}")
