"""Device memory before and after 200 cycles of net creation, two identify calls and net destruction (what a net
keeps between identify calls must go with it)."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import streamz_amd as sz
rng = np.random.default_rng(1)
pcm = np.clip(np.round(0.3 * rng.standard_normal(44100 * 30) * 32767), -32768, 32767).astype(np.int16)
ext = sz.FeatureExtractor()
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=1); sz.identify_speaker_list(net, pcm, 0.5, ext); del net
f0 = free()
for i in range(200):
    net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=i)
    a = sz.identify_speaker_list(net, pcm, 0.5, ext)
    b = sz.identify_speaker(net, pcm[:44100 * (1 + i % 7)], ext)
    del net
f1 = free()
print("free before", f0, "after", f1, "difference", f0 - f1)
assert abs(f0 - f1) < (64 << 20), "device memory is leaking"
print("ok")
