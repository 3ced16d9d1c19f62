"""Generate the golden fixtures under tests/golden/.

The reference is Rust and cannot be built or imported in the build container, and its own
tests hold no numeric vector for this path, so these fixtures do NOT come from the
reference.  They come from an independent float64 numpy/scipy formulation of the same
mathematics (np.fft.rfft, the librosa mel-filter formula, scipy DCT-II / 2, numpy
broadcasting for the network), written without looking at oracle/streamz_oracle.c's loops.
tests/test_oracle_golden.py checks the C oracle against them; the HIP path is then checked
against both.

Run:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz)
"""
import os

import numpy as np
import scipy.fft

HERE = os.path.dirname(os.path.abspath(__file__))
SR, WIN, HOP, NMEL, NMFCC = 44100, 800, 400, 26, 20


# ---------------------------------------------------------------- mel filterbank (Slaney)
def hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank():
    fftfreqs = np.linspace(0, SR / 2, WIN // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(SR / 2), NMEL + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((NMEL, WIN // 2 + 1))
    for i in range(NMEL):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:NMEL + 2] - mel_f[:NMEL])
    return w * enorm[:, None]


# ---------------------------------------------------------------- features
def deltas(x):
    if len(x) == 0:
        return x.copy()
    nxt = np.concatenate([x[1:], x[-1:]])
    prv = np.concatenate([x[:1], x[:-1]])
    return (nxt - prv) / 2.0


def features(pcm):
    pcm = np.asarray(pcm, np.int16)
    n = 0 if pcm.size < WIN else (pcm.size - WIN) // HOP + 1
    if n == 0:
        return np.zeros((0, 3 * NMFCC)), np.zeros((0, NMFCC))
    x = (pcm.astype(np.float32) / np.float32(32767.0)).astype(np.float64)
    idx = np.arange(WIN)[None, :] + HOP * np.arange(n)[:, None]
    spec = np.fft.rfft(x[idx], axis=1)
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ mel_filterbank().T
    logmel = np.log(np.maximum(mel, 1e-12))
    base = (scipy.fft.dct(logmel, type=2, axis=1) / 2.0)[:, :NMFCC]
    d1 = deltas(base)
    d2 = deltas(d1)
    f = np.concatenate([base, d1, d2], axis=1)
    mean = f.mean(axis=1, keepdims=True)
    std = np.maximum(np.sqrt(((f - mean) ** 2).mean(axis=1, keepdims=True)), 1e-6)
    return (f - mean) / std, base


# ---------------------------------------------------------------- network
def net_forward(p, x):
    w1, b1, w2, b2, w3, b3 = p
    h1pre = x @ w1 + b1
    h1 = np.where(h1pre > 0, h1pre, 0.0)
    h2 = np.tanh(h1 @ w2 + b2)
    z = h2 @ w3 + b3
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return h1pre, h1, h2, e / e.sum(axis=1, keepdims=True)


def net_grads(p, x, t):
    w1, b1, w2, b2, w3, b3 = p
    h1pre, h1, h2, o = net_forward(p, x)
    d3 = o - t
    d2 = (d3 @ w3.T) * (1 - h2 ** 2)
    d1 = (d2 @ w2.T) * (h1pre > 0)
    return [x.T @ d1, d1.sum(0), h1.T @ d2, d2.sum(0), h2.T @ d3, d3.sum(0)]


def train_batch(p, x, target, lr):
    if len(x) == 0:
        return [a.copy() for a in p]
    t = np.broadcast_to(target, (len(x), len(target)))
    g = net_grads(p, x, t)
    return [a - (lr / len(x)) * b for a, b in zip(p, g)]


def pretrain_from_features(p, wins, cls, ncls, epochs, lr, batch, perm, keep):
    target = np.zeros(ncls)
    if cls < ncls:
        target[cls] = 1.0
    total, count = 0.0, 0
    for ep in range(epochs):
        rows = wins[perm[ep]] * (keep[ep] != 0)
        for c0 in range(0, len(rows), max(batch, 1)):
            chunk = rows[c0:c0 + max(batch, 1)]
            live = chunk[np.any(chunk != 0.0, axis=1)]
            if len(live):
                o = net_forward(p, live)[3]
                total += float(-(target[None, :] * np.log(np.maximum(o, 1e-12))).sum())
                count += len(live)
            p = train_batch(p, live, target, lr)
    return p, (total / count if count else 0.0)


def identify(p, feats, thr):
    o = net_forward(p, feats)[3]
    C = o.shape[1]
    best = C - 1 - np.argmax(o[:, ::-1], axis=1)  # last maximum wins
    ok = o[np.arange(len(o)), best] >= thr
    counts = np.bincount(best[ok], minlength=C)
    ids = [i for i in range(C) if counts[i] > 0]
    ids.sort(key=lambda i: -counts[i])  # stable
    return np.array(ids, np.uint32), counts


# ---------------------------------------------------------------- inputs
def synth_clip(rng, n, kind):
    t = np.arange(n) / SR
    if kind == "noise":
        x = 0.3 * rng.standard_normal(n)
    elif kind == "tones":
        x = 0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 3520.0 * t + 0.3)
        x += 0.001 * rng.standard_normal(n)
    elif kind == "voiced":
        f0 = 110.0
        x = sum(a * np.sin(2 * np.pi * f0 * (h + 1) * t) for h, a in enumerate([0.2, 0.12, 0.08, 0.05]))
        x += 0.25 * 10 ** (-30 / 20) * rng.standard_normal(n)
    elif kind == "quiet":
        x = 3e-4 * rng.standard_normal(n)
    elif kind == "silence_then_loud":
        x = np.zeros(n)
        x[n // 2:] = 0.5 * rng.standard_normal(n - n // 2)
    elif kind == "fullscale":
        x = np.where(rng.random(n) < 0.5, -1.0001, 1.0)
    else:
        raise ValueError(kind)
    return np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)


def rand_net(rng, dims):
    i, h1, h2, o = dims
    return [rng.uniform(-0.5, 0.5, (i, h1)), rng.uniform(-0.1, 0.1, h1), rng.uniform(-0.5, 0.5, (h1, h2)),
            rng.uniform(-0.1, 0.1, h2), rng.uniform(-0.5, 0.5, (h2, o)), rng.uniform(-0.1, 0.1, o)]


def main():
    rng = np.random.default_rng(20240611)
    # ---- feature fixtures
    clips = {}
    for kind, n in [("noise", 8000), ("tones", 12345), ("voiced", 22050), ("quiet", 4000),
                    ("silence_then_loud", 6000), ("fullscale", 2000)]:
        clips[kind] = synth_clip(rng, n, kind)
    clips["short799"] = synth_clip(rng, 799, "noise")
    clips["exact800"] = synth_clip(rng, 800, "noise")
    clips["len1199"] = synth_clip(rng, 1199, "noise")
    clips["len1200"] = synth_clip(rng, 1200, "noise")
    clips["zeros"] = np.zeros(2400, np.int16)
    out = {"mel_filterbank": mel_filterbank().astype(np.float32)}
    for k, pcm in clips.items():
        f, b = features(pcm)
        out[f"pcm_{k}"] = pcm
        out[f"feat_{k}"] = f.astype(np.float32)
        out[f"base_{k}"] = b.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "features.npz"), **out)

    # ---- network fixtures
    net = {}
    for name, dims, B in [("tiny", (4, 3, 2, 2), 1), ("mid", (60, 32, 16, 4), 8), ("wide", (60, 128, 64, 3), 5)]:
        p = rand_net(rng, dims)
        x = rng.standard_normal((B, dims[0]))
        tgt = np.zeros(dims[3])
        tgt[dims[3] - 1] = 1.0
        probs = net_forward(p, x)[3]
        p2 = train_batch(p, x, tgt, 0.05)
        for i, a in enumerate(p):
            net[f"{name}_p{i}"] = a.astype(np.float32)
        for i, a in enumerate(p2):
            net[f"{name}_q{i}"] = a.astype(np.float32)
        net[f"{name}_x"] = x.astype(np.float32)
        net[f"{name}_target"] = tgt.astype(np.float32)
        net[f"{name}_probs"] = probs.astype(np.float32)
        net[f"{name}_dims"] = np.array(dims, np.int32)
    # epoch loop with injected shuffle / dropout, incl. all-zero rows and a short last chunk
    dims = (60, 24, 12, 3)
    p = rand_net(rng, dims)
    n_win, epochs, batch = 21, 3, 8
    wins = rng.standard_normal((n_win, 60))
    wins[5] = 0.0  # a window that is already all-zero is skipped (lib.rs:607-609)
    perm = np.stack([rng.permutation(n_win) for _ in range(epochs)]).astype(np.uint32)
    keep = (rng.random((epochs, n_win, 60)) >= 0.2).astype(np.uint8)
    keep[1, 8:16] = 0  # a whole chunk dropped -> empty batch -> no update (lib.rs:1003-1005)
    keep[2, 3] = 0
    pq, loss = pretrain_from_features([a.astype(np.float32).astype(np.float64) for a in p],
                                      wins.astype(np.float32).astype(np.float64), 1, 3, epochs, 0.05, batch,
                                      perm, keep)
    for i, a in enumerate(p):
        net[f"loop_p{i}"] = a.astype(np.float32)
    for i, a in enumerate(pq):
        net[f"loop_q{i}"] = a.astype(np.float32)
    net["loop_wins"] = wins.astype(np.float32)
    net["loop_perm"] = perm
    net["loop_keep"] = keep
    net["loop_loss"] = np.float32(loss)
    net["loop_dims"] = np.array(dims, np.int32)
    net["loop_cfg"] = np.array([1, 3, epochs, batch], np.int32)  # class, num_classes, epochs, batch
    # identify_speaker_list on feature rows, incl. an exact tie (last index wins) and a threshold miss
    dims = (60, 16, 8, 4)
    p = rand_net(rng, dims)
    p[4][:, 3] = p[4][:, 2]
    p[5][3] = p[5][2]  # classes 2 and 3 always tie
    feats = rng.standard_normal((40, 60))
    ids, counts = identify([a.astype(np.float32).astype(np.float64) for a in p],
                           feats.astype(np.float32).astype(np.float64), 0.3)
    for i, a in enumerate(p):
        net[f"ident_p{i}"] = a.astype(np.float32)
    net["ident_feats"] = feats.astype(np.float32)
    net["ident_ids"] = ids
    net["ident_counts"] = counts.astype(np.uint64)
    net["ident_dims"] = np.array(dims, np.int32)
    np.savez_compressed(os.path.join(HERE, "net.npz"), **net)
    print("wrote", os.path.join(HERE, "features.npz"), os.path.join(HERE, "net.npz"))


if __name__ == "__main__":
    main()
