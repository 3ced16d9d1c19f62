#!/usr/bin/env python3
"""Headline benchmark: feature-windows/sec (extract + train) on synthetic 44.1 kHz PCM.

One "step" = one pass of the hot path over the rank's clips, everything device-resident:
  extract   PCM (int16, already in HBM) -> 60 z-scored MFCC+delta+delta-delta features per window
  train     one epoch over all windows: forward, softmax/cross-entropy, backward, gradients summed
            on the device, [all-reduce of the 0.67 MB gradient vector across ranks,] one SGD update
windows/sec = windows of all ranks / (t_extract + t_epoch), max over ranks, K timed steps.

Default workload = BASELINE.json configs[1]: 1 000 clips x 60 s, 16 speakers, net 60-512-256-16.
--scaling strong (the default; the metric is quoted on one fixed set of 1 000 clips at 1/2/4/8 GPUs): the
clips are cut into contiguous ranges balanced on window count (szh_shard_ranges), one range per rank.
--scaling weak: every rank holds its own --clips clips.  Either way there is no data-path collective
except the per-epoch gradient all-reduce.

Launch:  python bench.py --gpus 1 --steps 3 --warmup 1
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # before streamz_amd: one HIP runtime per process

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import streamz_amd as sz  # noqa: E402

SR = 44100
PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X dense f32 MFMA / vector peak
PEAK_BF16_MATRIX_TFLOPS = 2516.6  # dense bf16 MFMA: 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz
X3_KERNELS = ("fwd2", "delta1", "dw2")  # the products --gemm bf16x3 moves to the bf16 matrix cores, six MFMAs per term
PEAK_HBM_GBPS = 8000.0


class _DevView:
    """Zero-copy torch view of library-owned device memory."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def synth_clips(n_clips, seconds, speakers, device, first_clip=0):
    """Speaker s = 4 harmonics of f0 = 90 + 12 s Hz + white noise at -30 dB, about 0.25 FS."""
    n = int(seconds * SR)
    n_al = (n + 7) // 8 * 8
    pcm = torch.zeros(n_clips * n_al, dtype=torch.int16, device=device)
    t = torch.arange(n, device=device, dtype=torch.float64) / SR
    amps = [1.0, 0.6, 0.4, 0.25]
    gen = torch.Generator(device=device)
    for i in range(n_clips):
        clip = first_clip + i
        s = clip % speakers
        f0 = 90.0 + 12.0 * s
        x = torch.zeros(n, device=device, dtype=torch.float64)
        for h, a in enumerate(amps):
            a_s = a * (1.0 + 0.3 * np.sin(1.7 * s + h))  # per-speaker formant-ish pattern
            x += a_s * torch.sin(2 * np.pi * f0 * (h + 1) * t + 0.5 * h)
        x *= 0.25 / float(sum(amps))
        gen.manual_seed(1234 + clip)
        x += 0.25 * 10 ** (-30 / 20) * torch.randn(n, device=device, dtype=torch.float32, generator=gen).double()
        pcm[i * n_al:i * n_al + n] = torch.clamp(torch.round(x * 32767.0), -32768, 32767).to(torch.int16)
    offsets = np.arange(n_clips, dtype=np.uint64) * n_al
    lengths = np.full(n_clips, n, np.uint64)
    return pcm, offsets, lengths


def cpu_baseline(speakers, hidden, seconds):
    """The CPU restatement (oracle/) on a bounded sample: extraction with OpenMP over clips as
    main.rs:500-508 does, training serial at batch 8 as lib.rs:599-622 does."""
    from oracle import pyoracle as po
    threads = min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    n_clips = 32
    pcm, off, ln = synth_clips(n_clips, seconds, speakers, "cpu")
    pcm = pcm.numpy()
    ext = po.Extractor()
    ext.extract_files(pcm[: int(ln[0])], off[:1], ln[:1], fft_mode=1)  # warm tables / threads
    t0 = time.perf_counter()
    feats, woff, nw = ext.extract_files(pcm, off, ln, fft_mode=1)
    t_ext = time.perf_counter() - t0
    n_ext = int(nw.sum())
    net = po.Net(60, hidden[0], hidden[1], speakers)
    rng = np.random.default_rng(0)
    net.set_params([rng.uniform(-0.5, 0.5, s).astype(np.float32) if len(s) == 2 else np.zeros(s, np.float32)
                    for s in net.shapes()])
    n_train = 30000
    t0 = time.perf_counter()
    net.pretrain_from_features(feats[:n_train], 1, speakers, 1, 0.01, 0.2, 8)
    t_tr = time.perf_counter() - t0
    per_win = t_ext / n_ext + t_tr / n_train
    # The schedule the headline runs (one optimiser step per epoch, gradients summed over all rows),
    # on all host threads: each thread sums the gradient of a slice of rows (ctypes releases the GIL).
    from concurrent.futures import ThreadPoolExecutor
    n_batched = 4000 * threads
    fb = np.ascontiguousarray(np.resize(feats, (n_batched, 60)))
    lb = (np.arange(n_batched) % speakers).astype(np.int32)
    sl = np.array_split(np.arange(n_batched), threads)
    with ThreadPoolExecutor(threads) as pool:
        t0 = time.perf_counter()
        parts = list(pool.map(lambda ix: net.grad(fb[ix[0]:ix[-1] + 1], labels=lb[ix[0]:ix[-1] + 1]), sl))
        net.apply(np.sum(parts, axis=0), 0.01 / n_batched)
        t_bt = time.perf_counter() - t0
    return {
        "value": 1.0 / per_win,
        "unit": "windows/s",
        "cores": threads,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": (f"extract: {n_clips} clips x {seconds:g} s = {n_ext} windows on {threads} OpenMP threads "
                   f"({n_ext / t_ext:.0f} windows/s); train: one epoch of {n_train} windows at batch 8, dropout 0.2, "
                   f"serial on 1 core as the reference does ({n_train / t_tr:.0f} windows/s); "
                   "value = 1 / (s_per_window_extract + s_per_window_train)"),
        "extract_windows_per_s": n_ext / t_ext,
        "train_windows_per_s": n_train / t_tr,
        "batched_train_windows_per_s": n_batched / t_bt,
        "batched_value": 1.0 / (t_ext / n_ext + t_bt / n_batched),
        "batched_note": (f"the headline's schedule on the CPU: gradients of {n_batched} windows summed by {threads} threads, "
                         "one update; batched_value = 1 / (s_per_window_extract + s_per_window_batched_train)"),
    }


def measured_traffic(kernel, rows_per_launch, gemm="f32"):
    """HBM bytes per launch of `kernel` from the newest committed FETCH_SIZE / WRITE_SIZE passes of the
    same GEMM mode (profiles/*_<mode>_pmc_traffic.json: separate rocprofv3 --pmc runs of
    tools/net_bench on the same kernels and row counts, FETCH_SIZE doubled as gfx950 requires), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{gemm}_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    k = t.get("kernels", {}).get(kernel)
    if not k:
        return None, None
    return k["hbm_bytes_per_row"] * rows_per_launch, os.path.relpath(files[-1], ROOT)


def other_figures(sz, ext, plan, pcm, feats, net_dims, n_win, dev, lr, dropout):
    """Rates reported beside the headline: inference (extract + per-window vote, the
    identify_speaker_list path) over the same clips, and the reference's own training schedule
    (batch 8, one optimiser step per 8 windows, lib.rs:599-622) on one clip's windows."""
    stream = torch.cuda.current_stream().cuda_stream
    net = sz.SimpleNeuralNet(*net_dims, seed=7)
    counts = torch.zeros(net_dims[3], dtype=torch.int64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(2):
        if it == 1:
            ev[0].record()
        plan.run(pcm.data_ptr(), feats.data_ptr(), stream)
        net.vote_dev(feats.data_ptr(), n_win, 0.5, counts.data_ptr(), stream)
    ev[1].record()
    torch.cuda.synchronize()
    t_inf = ev[0].elapsed_time(ev[1]) * 1e-3
    n_b8 = min(n_win, 6614)
    host = feats[:n_b8].cpu().numpy()
    sz.pretrain_from_features(net, host[:64], 1, net_dims[3], 1, lr, dropout, 8, seed=1)  # warm
    b8_epochs = 10
    t0 = time.perf_counter()
    sz.pretrain_from_features(net, host, 1, net_dims[3], b8_epochs, lr, dropout, 8, seed=2)
    t_b8 = (time.perf_counter() - t0) / b8_epochs
    # the inference-only configuration: a 100 h recording (39 689 999 windows) already in device
    # memory, identify_speaker_list in tiles (262144 windows by default) replayed from a HIP graph
    n_stream = 100 * 3600 * 44100
    block = pcm[:400 * 3000]
    rec = block.repeat((n_stream + block.numel() - 1) // block.numel())[:n_stream].contiguous()
    torch.cuda.synchronize()
    t_stream = None
    for _ in range(3):
        t0 = time.perf_counter()
        sz.identify_speaker_list_dev(net, rec.data_ptr(), n_stream, 0.5, ext)
        dt = time.perf_counter() - t0
        t_stream = dt if t_stream is None else min(t_stream, dt)
    # the same call as the Rust host makes it, on a slice in pageable host memory (10 h: 3.2 GB): bound by the copy
    n_host = 10 * 3600 * 44100
    rec_host = rec[:n_host].cpu().numpy()
    del rec
    t_host = None
    for _ in range(2):
        t0 = time.perf_counter()
        sz.identify_speaker_list(net, rec_host, 0.5, ext)
        dt = time.perf_counter() - t0
        t_host = dt if t_host is None else min(t_host, dt)
    del rec_host
    # flops of one window's forward pass (lib.rs:880-891): the three layers' multiply-adds
    fwd_flops = 2 * (net_dims[0] * net_dims[1] + net_dims[1] * net_dims[2] + net_dims[2] * net_dims[3])

    def mfma_roofline(windows_per_s, what):
        ach = fwd_flops * windows_per_s / 1e12
        return {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MATRIX_TFLOPS,
                "flops_per_window": fwd_flops, "note": what}
    return {"identify_100h_windows_per_s": sz.window_count(n_stream) / t_stream,
            "identify_100h_roofline": mfma_roofline(sz.window_count(n_stream) / t_stream,
                                                    "whole call (extraction of every tile included) against the f32 matrix peak"),
            "infer_roofline": mfma_roofline(n_win / t_inf, "extract + vote over all clips against the f32 matrix peak"),
            "identify_100h_seconds": t_stream,
            "identify_100h_note": "configs[3]: identify_speaker_list over 100 h of PCM resident in HBM, tiles of 262144 "
                                  "windows, one HIP graph replay per tile, votes read once at the end",
            "identify_10h_from_host_windows_per_s": sz.window_count(n_host) / t_host,
            "identify_10h_from_host_note": "identify_speaker_list on 10 h of PCM in pageable host memory, the PCIe copy inside the call "
                                           "(%.1f GB/s of PCM; staged through pinned memory by helper threads)" % (n_host * 2 / t_host / 1e9),
            "infer_windows_per_s": n_win / t_inf,
            "infer_note": "extract + forward + thresholded vote (lib.rs:1383-1411) over all clips, device-resident",
            "cli_batch8_train_windows_per_s": n_b8 / t_b8,
            "cli_batch8_note": f"pretrain_from_features, batch 8, dropout {dropout:g}, {b8_epochs} epochs of {n_b8} windows in one call "
                               "through the host API (the CLI runs 100 per file), wall time of the call"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=1000, help="clips of the whole job (--scaling strong) or per GPU (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --clips clips in all, sharded over the ranks by window count; weak: --clips clips per rank")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--speakers", type=int, default=16)
    ap.add_argument("--hidden1", type=int, default=512)
    ap.add_argument("--hidden2", type=int, default=256)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--gemm", choices=["f32", "bf16x3"], default="f32",
                    help="arithmetic of the three large products: f32 matrix instructions, or f32 operands split "
                         "exactly into three bf16 pieces on the bf16 matrix cores (f32 accumulate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal of the N > 1 path on fewer GPUs than ranks: "
                    "ranks share a GPU and the library's communicator sums on the host (shared memory) instead of by RCCL")
    ap.add_argument("--no-extras", action="store_true", help="skip the inference and batch-8 side measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_FORCE_DIST=1 takes the multi-rank path with a single rank (rehearsal of the collective on one GPU)
    multi = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    if multi:
        # torch.distributed is the control plane only (rendezvous, barriers, the max over ranks of the
        # elapsed time): a gloo group over the loopback interface.  The data-path collective is the
        # library's own RCCL all-reduce on the gradient vector (szh_net_allreduce_grad_dev).
        import torch.distributed as dist
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        # RCCL's bootstrap (rendezvous of the ranks' communicators) is a socket exchange too; all ranks are on
        # this node, so the loopback interface always reaches them (the data path is xGMI / device memory)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if args.share_gpu:
            local_rank %= torch.cuda.device_count()  # rehearsal: ranks share a GPU, RCCL would refuse them
            os.environ["STREAMZ_COMM"] = "host"      # the library's communicator then sums in shared memory
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo")
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if multi else 0)
    sz._capi._check(sz.lib().szh_init(dev.index))
    comm, torch_rccl, comm_note = None, None, None
    if multi:
        err = None
        try:
            if os.environ.get("BENCH_FORCE_TORCH_RCCL"):
                raise RuntimeError("BENCH_FORCE_TORCH_RCCL is set")
            ident = [sz.comm_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001 - whatever stops the library's communicator from coming up
            ident, err = [None], repr(e)
        dist.broadcast_object_list(ident, src=0)
        if ident[0] is not None:
            try:
                comm = sz.Comm(ident[0], rank, world)  # ncclCommInitRank: collective over the ranks
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        # every rank takes the same collective: if the library's communicator is missing anywhere, all
        # ranks reduce the same library-owned gradient vector in place through torch's RCCL group instead
        ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm = None
            if args.share_gpu:
                raise SystemExit("bench: the library's host communicator did not come up: %s" % err)
            torch_rccl = dist.new_group(backend="nccl")
            comm_note = "library communicator unavailable (%s): torch.distributed RCCL group on the same vector" % err
            if rank == 0:
                print("bench: " + comm_note, file=sys.stderr)

    # ---- synthetic, device-resident inputs
    if args.scaling == "strong":  # one fixed set of clips, cut over the ranks by window count
        per_clip = sz.window_count(int(args.seconds * SR))
        cut = sz.shard_ranges([per_clip] * args.clips, world)
        first_clip, my_clips = cut[rank], cut[rank + 1] - cut[rank]
    else:
        first_clip, my_clips = rank * args.clips, args.clips
    pcm, offsets, lengths = synth_clips(my_clips, args.seconds, args.speakers, dev, first_clip=first_clip)
    ext = sz.FeatureExtractor()
    plan = sz.Plan(ext, offsets, lengths)
    n_win = plan.total_windows
    feats = torch.empty((n_win, 60), dtype=torch.float32, device=dev)
    per_file = np.diff(plan.window_offsets).astype(np.int64)
    clip_speaker = (np.arange(my_clips) + first_clip) % args.speakers
    labels = torch.from_numpy(np.repeat(clip_speaker, per_file).astype(np.int32)).to(dev)
    net = sz.SimpleNeuralNet(60, args.hidden1, args.hidden2, args.speakers, seed=42)  # same init on every rank
    net.set_gemm(args.gemm)
    grad = torch.as_tensor(_DevView(net.grad_ptr(), net.grad_len()), device=dev)
    if comm is not None:
        comm.broadcast_params(net, 0, torch.cuda.current_stream().cuda_stream)
    stream = torch.cuda.current_stream().cuda_stream

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step(k, record=False):
        if record:
            ev[0].record()
        if n_win:
            plan.run(pcm.data_ptr(), feats.data_ptr(), stream)
        if record:
            ev[1].record()
        net.zero_grad_dev(stream)
        net.accumulate_dev(feats.data_ptr(), labels.data_ptr(), n_win, args.dropout, 1000 + k, stream)
        if comm is not None:
            comm.allreduce_grad(net, stream)  # gradients, loss and live-row count summed over ranks, in place
        elif torch_rccl is not None:
            dist.all_reduce(grad, group=torch_rccl)
        net.apply_grad_dev(args.lr, stream)
        if record:
            ev[2].record()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(first_k, warmup, steps):
        """warmup untimed steps, then `steps` timed ones between barriers; per-kernel event times of the timed ones"""
        for k in range(warmup):
            step(first_k + k)
        barrier()
        if not args.no_profile:
            sz.profile_enable(True)
            sz.profile_report()
        t0 = time.perf_counter()
        for k in range(steps):
            step(first_k + warmup + k, record=(k == steps - 1))
        barrier()
        dt = time.perf_counter() - t0
        pr = {} if args.no_profile else sz.profile_report()
        sz.profile_enable(False)
        ls, lv = [float(v) for v in grad[-2:].tolist()]
        return dt, pr, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), (ls / lv if lv > 0 else None)

    elapsed, prof, t_ext_ms, t_train_ms, mean_loss = timed(0, args.warmup, args.steps)

    windows_by_rank = [n_win]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        gathered = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([n_win], dtype=torch.int64))
        windows_by_rank = [int(t.item()) for t in gathered]
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = 1000.0 * elapsed / args.steps
    total_windows = sum(windows_by_rank)
    value = total_windows * args.steps / elapsed

    # ---- roofline of the dominant kernel, from the event pairs recorded during the timed steps
    C, H1, H2 = args.speakers, args.hidden1, args.hidden2
    flops_per_row = {"fwd1": 2 * 60 * H1, "fwd2": 2 * H1 * H2, "fwd3": 2 * H2 * C, "delta2": 2 * C * H2,
                     "delta1": 2 * H2 * H1, "dw3": 2 * H2 * C, "dw2": 2 * H1 * H2, "dw1": 2 * 60 * H1}
    roofline = None
    kernels = {}
    if prof:
        for tag, (cnt, ms) in prof.items():
            kernels[tag] = {"launches": cnt, "total_ms": round(ms, 3), "avg_us": round(1000.0 * ms / cnt, 2)}
        dom = max(prof, key=lambda k: prof[k][1])
        cnt, ms = prof[dom]
        if dom in flops_per_row:
            # algorithmic flops of all launches of this kernel in the timed steps / their total duration
            total_flops = flops_per_row[dom] * n_win * args.steps
            achieved = total_flops / (ms * 1e-3) / 1e12
            # a split-bf16 product spends six bf16 matrix instructions per algorithmic multiply-add
            peak = PEAK_BF16_MATRIX_TFLOPS / 6 if (args.gemm == "bf16x3" and dom in X3_KERNELS) else PEAK_F32_MATRIX_TFLOPS
            roofline = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": peak,
                        "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                        "avg_launch_us": 1000.0 * ms / cnt,
                        "launches": cnt, "flops_per_launch": total_flops / cnt,
                        "rows_per_launch": n_win * args.steps / cnt}
            roofline["traffic"], roofline["traffic_source"] = measured_traffic(dom, n_win * args.steps / cnt, args.gemm)
        else:
            total_bytes = 1040.0 * n_win * args.steps
            achieved = total_bytes / (ms * 1e-3) / 1e9
            roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBPS, "traffic": None, "avg_launch_us": 1000.0 * ms / cnt}
        ext_ms = sum(prof[t][1] for t in ("mfcc_base", "finish") if t in prof)
        if ext_ms > 0:
            ext_gbps = 1040.0 * n_win * args.steps / (ext_ms * 1e-3) / 1e9
            kernels["extract_total"] = {"algorithmic_GBps": round(ext_gbps, 1), "frac_of_hbm_peak": ext_gbps / PEAK_HBM_GBPS,
                                        "windows_per_s": n_win * args.steps / (ext_ms * 1e-3)}

    out = {
        "metric": "feature-windows/sec (extract+train)",
        "value": value,
        "unit": "windows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32" if args.gemm == "f32" else "f32 values as three bf16 pieces each on the bf16 matrix cores, f32 accumulate",
        "data": "synthetic",
        "config": {
            "workload": (f"configs[1]: {args.clips} x {args.seconds:g} s synthetic 44.1 kHz mono clips "
                         + ("in all, cut over the GPUs by window count" if args.scaling == "strong" else "per GPU")
                         + f", {args.speakers} speakers, net 60-{H1}-{H2}-{C}"),
            "scaling_conventions": "strong: --clips is the whole job's clip count (default); weak: --clips per GPU (--scaling weak)",
            "windows_per_gpu": n_win,
            "windows_by_rank": windows_by_rank,
            "windows_total": total_windows,
            "batch": "one optimiser step per epoch over all windows (gradients summed on device"
                     + (", one in-place all-reduce of the gradient vector across ranks)"
                        if (comm is not None or torch_rccl is not None) else ")"),
            "epochs_per_step": 1,
            "dropout": args.dropout,
            "lr": args.lr,
            "parallelism": f"files sharded over {world} GPU(s)",
            "collective": ("none (one rank)" if not multi else "library host transport (ranks share a GPU: rehearsal)" if args.share_gpu
                           else "library RCCL communicator, in place on the gradient vector" if comm is not None
                           else "torch.distributed RCCL group (fallback)"),
            **({"collective_note": comm_note} if comm_note else {}),
        },
        "breakdown_ms": {"extract": t_ext_ms, "train_epoch": t_train_ms},
        "extract_windows_per_s": n_win / (t_ext_ms * 1e-3),
        "train_windows_per_s": n_win / (t_train_ms * 1e-3),
        "mean_loss_last_step": mean_loss,
        "roofline": roofline,
        "kernels": kernels,
    }
    if world == 1 and not args.no_extras and args.gemm == "f32" and H1 % 256 == 0 and H2 % 256 == 0:
        # the same steps with the three large products on the split-bf16 kernels (opt-in; not the headline)
        net.set_gemm("bf16x3")
        dt, pr, te, tt, ml = timed(args.warmup + args.steps, 1, args.steps)
        net.set_gemm("f32")
        x3 = {"value": n_win * args.steps / dt, "unit": "windows/s", "ms_per_step": 1000.0 * dt / args.steps,
              "breakdown_ms": {"extract": te, "train_epoch": tt}, "mean_loss_last_step": ml,
              "arithmetic": "every f32 operand of fwd2 / delta1 / dw2 split exactly into three bf16 values; six bf16 "
                            "MFMAs per product term (the three dropped terms are below 2^-24 relative), f32 accumulate",
              "kernels": {t: {"avg_us": round(1000.0 * pr[t][1] / pr[t][0], 2),
                              "TFLOPs_f32_equivalent": round(flops_per_row[t] * n_win * args.steps / (pr[t][1] * 1e-3) / 1e12, 1),
                              "frac_of_bf16_peak_over_6": flops_per_row[t] * n_win * args.steps / (pr[t][1] * 1e-3) / 1e12
                                                          / (PEAK_BF16_MATRIX_TFLOPS / 6)}
                          for t in X3_KERNELS if t in pr}}
        out["bf16x3_variant"] = x3
    if world == 1 and not args.no_extras:
        out["other_figures"] = other_figures(sz, ext, plan, pcm, feats, (60, H1, H2, C), n_win, dev, args.lr, args.dropout)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.speakers, (H1, H2), args.seconds)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
