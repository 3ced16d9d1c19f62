"""Per-call time of FeatureExtractor.extract (host PCM in, host features out) for short clips."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import streamz_amd as sz
ext = sz.FeatureExtractor()
rng = np.random.default_rng(3)
for secs in ((1,) if os.environ.get("ONLY") else (1, 10, 60)):
    pcm = np.clip(np.round(0.3 * rng.standard_normal(int(secs * 44100)) * 32767), -32768, 32767).astype(np.int16)
    for _ in range(3): ext.extract(pcm)
    t0 = time.perf_counter(); n = 50
    for _ in range(n): ext.extract(pcm)
    dt = (time.perf_counter() - t0) / n
    print(f"{secs:4d} s clip ({sz.window_count(pcm.size)} windows): {dt*1e6:8.1f} us per extract call", flush=True)
