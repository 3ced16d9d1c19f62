"""identify_speaker_list over 100 h of device-resident PCM in tiles of TILES (default 8192,32768) windows, with the number
of tiles in flight taken from STREAMZ_TILE_LANES (1..4); the votes must not depend on it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import streamz_amd as sz
n_samples = int(100 * 3600 * 44100)
nw = sz.window_count(n_samples)
rng = np.random.default_rng(3)
block = np.clip(np.round(0.3 * rng.standard_normal(400 * 3000) * 32767), -32768, 32767).astype(np.int16)
dblock = torch.from_numpy(block).cuda()
stream = dblock.repeat((n_samples + block.size - 1) // block.size)[:n_samples].contiguous()
torch.cuda.synchronize()
net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=7)
ext = sz.FeatureExtractor()
ref = None
for tile in [int(t) for t in os.environ.get("TILES", "8192,32768").split(",")]:
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        ids, counts = sz.identify_speaker_list_dev(net, stream.data_ptr(), n_samples, 0.5, ext, tile_windows=tile)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    if ref is None: ref = counts
    assert (np.asarray(counts) == np.asarray(ref)).all(), (counts, ref)
    print("lanes", os.environ.get("STREAMZ_TILE_LANES"), "tile", tile, "seconds %.4f" % best, "M windows/s %.1f" % (nw / best / 1e6), flush=True)
print(list(ref))
