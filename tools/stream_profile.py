"""Short device-resident identify run for rocprofv3: python tools/stream_profile.py [hours] [tile]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import streamz_amd as sz  # noqa: E402

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n = int(hours * 3600 * 44100)
rng = np.random.default_rng(3)
block = torch.from_numpy(np.clip(np.round(0.3 * rng.standard_normal(400 * 3000) * 32767), -32768, 32767).astype(np.int16)).cuda()
stream = block.repeat((n + block.numel() - 1) // block.numel())[:n].contiguous()
torch.cuda.synchronize()
net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=7)
ext = sz.FeatureExtractor()
for _ in range(2):
    ids, counts = sz.identify_speaker_list_dev(net, stream.data_ptr(), n, 0.5, ext, tile_windows=tile)
print(ids, int(counts.sum()))
