"""Times identify_speaker_list over a long recording: device-resident PCM at several tile
sizes, and host PCM (copies included).  usage: python tools/stream_timing.py [hours] [classes]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import streamz_amd as sz  # noqa: E402


def main():
    hours = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    classes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    n_samples = int(hours * 3600 * 44100)
    nw = sz.window_count(n_samples)
    rng = np.random.default_rng(3)
    block = np.clip(np.round(0.3 * rng.standard_normal(400 * 3000) * 32767), -32768, 32767).astype(np.int16)
    dblock = torch.from_numpy(block).cuda()
    reps = (n_samples + block.size - 1) // block.size
    stream = dblock.repeat(reps)[:n_samples].contiguous()
    torch.cuda.synchronize()
    net = sz.SimpleNeuralNet(60, 512, 256, classes, seed=7)
    ext = sz.FeatureExtractor()
    out = {"hours": hours, "windows": nw, "classes": classes, "device_resident": {}, "host_pcm": {}}
    for tile in (8192, 32768, 131072, 393216):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            ids, counts = sz.identify_speaker_list_dev(net, stream.data_ptr(), n_samples, 0.5, ext, tile_windows=tile)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["device_resident"][str(tile)] = {"seconds": best, "windows_per_s": nw / best,
                                             "us_per_tile": best / max(1, nw // tile) * 1e6}
        print(tile, best, nw / best, flush=True)
    # host PCM: at most 10 h, pageable numpy memory as a caller of the reference API would hold it
    h_hours = min(hours, 10.0)
    h_samples = int(h_hours * 3600 * 44100)
    host = np.tile(block, (h_samples + block.size - 1) // block.size)[:h_samples].copy()
    for tile in ("8192", "131072"):
        os.environ["STREAMZ_TILE_WINDOWS"] = tile
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            sz.identify_speaker_list(net, host, 0.5, ext)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        hw = sz.window_count(h_samples)
        out["host_pcm"][tile] = {"hours": h_hours, "seconds": best, "windows_per_s": hw / best,
                                 "pcm_GB_per_s": h_samples * 2 / best / 1e9}
        print("host", tile, best, hw / best, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
