"""szh_net_vote_dev (forward + softmax + thresholded vote of feature rows already in HBM) at several batch
sizes; STREAMZ_NO_FUSED_VOTE=1 selects the per-layer kernels.  usage: python tools/vote_timing.py [classes]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import streamz_amd as sz
classes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = sz.SimpleNeuralNet(60, 512, 256, classes, seed=7)
flop = 2 * (60 * 512 + 512 * 256 + 256 * classes)
s = torch.cuda.current_stream().cuda_stream
for n in [int(v) for v in os.environ.get("ROWS", "8192,16384,32768,65536,262144,6614000").split(",")]:
    x = torch.randn((n, 60), device="cuda")
    counts = torch.zeros(classes, dtype=torch.int64, device="cuda")
    for _ in range(3):
        net.vote_dev(x.data_ptr(), n, 0.1, counts.data_ptr(), s)
    torch.cuda.synchronize()
    reps = max(3, min(200, int(2e8 / n)))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        net.vote_dev(x.data_ptr(), n, 0.1, counts.data_ptr(), s)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    print(f"fused={'0' if os.environ.get('STREAMZ_NO_FUSED_VOTE') else '1'} rows {n}: {us:.1f} us per call, {n / us:.1f} M rows/s, "
          f"{flop * n / us / 1e6:.1f} TFLOP/s ({flop * n / us / 1e6 / 157.3:.2f} of the f32 matrix peak)", flush=True)
    del x
