import os, sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import streamz_amd as sz
net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=7)
ext = sz.FeatureExtractor()
rng = np.random.default_rng(3)
for secs in (tuple(int(v) for v in os.environ["ONLY"].split(",")) if os.environ.get("ONLY") else (1, 10, 60, 600)):
    pcm = np.clip(np.round(0.3 * rng.standard_normal(int(secs * 44100)) * 32767), -32768, 32767).astype(np.int16)
    for _ in range(3): sz.identify_speaker_list(net, pcm, 0.5, ext)
    t0 = time.perf_counter(); n = 50
    for _ in range(n): sz.identify_speaker_list(net, pcm, 0.5, ext)
    dt = (time.perf_counter() - t0) / n
    print(f"{secs:4d} s clip ({sz.window_count(pcm.size)} windows): {dt*1e6:8.1f} us per identify_speaker_list call from host memory", flush=True)
