"""Latency of the host entry points a Rust shim calls per window or per batch:
szh_net_forward with one row (SimpleNeuralNet::forward, lib.rs:880) and szh_net_train_batch with
eight rows (lib.rs:1002).  Prints microseconds per call (median of batches of calls)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import streamz_amd as sz

rng = np.random.default_rng(0)
net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=3)
x1 = rng.standard_normal((1, 60)).astype(np.float32)
x8 = rng.standard_normal((8, 60)).astype(np.float32)
t = np.zeros(16, np.float32)
t[3] = 1.0
out = {}
for name, fn, reps in (("forward_B1", lambda: net.forward(x1), 300), ("forward_B8", lambda: net.forward(x8), 300),
                       ("train_batch_B8", lambda: net.train_batch(x8, t, 0.01), 200)):
    for _ in range(20):
        fn()
    samples = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        samples.append((time.perf_counter() - t0) / reps * 1e6)
    out[name + "_us"] = round(float(np.median(samples)), 1)
print(json.dumps(out))
