"""identify_speaker_list on a long recording in pageable HOST memory (HOURS, default 10 h of 44.1 kHz PCM): the call
is bound by the copy to the device.  STREAMZ_COPY_THREADS=0 takes the plain pageable hipMemcpyAsync per tile, the
default stages the copy through pinned memory with helper threads; STREAMZ_TILE_WINDOWS sets the tile.  The order
of speakers is checked against the device-resident call's."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import streamz_amd as sz
hours = float(os.environ.get("HOURS", "10"))
n_samples = int(hours * 3600 * 44100)
nw = sz.window_count(n_samples)
rng = np.random.default_rng(3)
block = np.clip(np.round(0.3 * rng.standard_normal(400 * 3000) * 32767), -32768, 32767).astype(np.int16)
pcm = np.tile(block, (n_samples + block.size - 1) // block.size)[:n_samples].copy()
net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=7)
ext = sz.FeatureExtractor()
dev = torch.from_numpy(pcm).cuda()
want, _ = sz.identify_speaker_list_dev(net, dev.data_ptr(), n_samples, 0.5, ext)
del dev
best = None
for _ in range(3):
    t0 = time.perf_counter()
    ids = sz.identify_speaker_list(net, pcm, 0.5, ext)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
    assert list(ids) == list(want), (ids, want)
print("copy threads %s tile %s: %.1f h from host memory in %.4f s = %.1f M windows/s, %.1f GB/s of PCM" % (
    os.environ.get("STREAMZ_COPY_THREADS", "default"), os.environ.get("STREAMZ_TILE_WINDOWS", "default"), hours, best,
    nw / best / 1e6, n_samples * 2 / best / 1e9), flush=True)
