"""Device memory, host memory and thread count before and after cycles of net creation, identify calls and net
destruction (what a net keeps between identify calls must go with it; the helper threads and the pinned ring of a
long host recording's staged copy must be gone after the call / with the net), and after repeated large
extract_batch calls."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import streamz_amd as sz
rng = np.random.default_rng(1)
pcm = np.clip(np.round(0.3 * rng.standard_normal(44100 * 30) * 32767), -32768, 32767).astype(np.int16)
long_pcm = np.tile(pcm, 16)   # 42 MB: staged copy, one piece
ext = sz.FeatureExtractor()
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
def status(key):
    for line in open("/proc/self/status"):
        if line.startswith(key + ":"): return int(line.split()[1])
def cycle(i, big):
    net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=i)
    sz.identify_speaker_list(net, long_pcm if big else pcm, 0.5, ext)
    sz.identify_speaker(net, pcm[:44100 * (1 + i % 7)], ext)
    del net
clips = [long_pcm[: 44100 * 60 * 2 + 17 * k] for k in range(8)]   # 84 MB in all
cycle(0, False); cycle(0, True); ext.extract_batch(clips)
f0, rss0, th0 = free(), status("VmRSS"), status("Threads")
for i in range(200): cycle(i, False)
for i in range(30): cycle(i, True)
for i in range(10): ext.extract_batch(clips)
f1, rss1, th1 = free(), status("VmRSS"), status("Threads")
print("device memory free before", f0, "after", f1, "difference", f0 - f1)
print("host resident set before %d kB after %d kB; threads before %d after %d" % (rss0, rss1, th0, th1))
assert abs(f0 - f1) < (64 << 20), "device memory is leaking"
assert rss1 - rss0 < (256 << 10), "host memory is leaking"
assert th1 == th0, "helper threads left behind"
print("ok")
