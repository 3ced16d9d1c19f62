"""ctypes binding of the CPU oracle (oracle/streamz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped package (streamz_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libstreamz_oracle.so")

FEAT = 60
MFCC = 20
NMELS = 26
NBINS = 401
WINDOW = 800
HOP = 400


def build(force=False):
    """Compile the oracle with gcc if the shared object is missing (or force=True)."""
    src = os.path.join(_HERE, "streamz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u64, u32, f32, vp = C.c_uint64, C.c_uint32, C.c_float, C.c_void_p
        L.so_window_count.restype = u64
        L.so_window_count.argtypes = [u64]
        L.so_mel_filterbank.argtypes = [vp]
        L.so_extractor_create.restype = vp
        L.so_extractor_destroy.argtypes = [vp]
        L.so_extract.restype = u64
        L.so_extract.argtypes = [vp, vp, u64, C.c_int, vp, vp]
        L.so_extract_files.argtypes = [vp, vp, vp, vp, u64, C.c_int, vp, vp]
        L.so_finish_features.argtypes = [vp, u64, vp]
        L.so_net_create.restype = vp
        L.so_net_create.argtypes = [C.c_int] * 4
        L.so_net_destroy.argtypes = [vp]
        L.so_net_param.restype = C.POINTER(C.c_float)
        L.so_net_param.argtypes = [vp, C.c_int]
        L.so_net_param_count.restype = C.c_size_t
        L.so_net_param_count.argtypes = [vp]
        L.so_net_forward.argtypes = [vp, vp, u64, vp]
        L.so_net_embed.argtypes = [vp, vp, u64, vp]
        L.so_net_forward_embedding.argtypes = [vp, vp, u64, vp]
        L.so_net_grad.argtypes = [vp, vp, u64, vp, vp, vp]
        L.so_net_apply.argtypes = [vp, vp, f32]
        L.so_net_train_batch.argtypes = [vp, vp, u64, vp, f32]
        L.so_net_train.argtypes = [vp, vp, vp, f32]
        L.so_seed.argtypes = [u64]
        L.so_augment.argtypes = [vp, u64, f32, u64, vp, vp]
        L.so_pretrain_from_features.restype = f32
        L.so_pretrain_from_features.argtypes = [vp, vp, u64, u32, u32, u32, f32, f32, u32, vp, vp]
        L.so_identify_from_features.restype = u32
        L.so_identify_from_features.argtypes = [vp, vp, u64, f32, vp, vp]
        L.so_identify_speaker_feats.restype = u32
        L.so_identify_speaker_feats.argtypes = [vp, vp, u64, vp]
        L.so_identify_speaker_with_threshold_feats.restype = C.c_int64
        L.so_identify_speaker_with_threshold_feats.argtypes = [vp, vp, u64, f32]
        L.so_identify_speaker_list.restype = u32
        L.so_identify_speaker_list.argtypes = [vp, vp, vp, u64, f32, C.c_int, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def window_count(n_samples):
    return int(lib().so_window_count(int(n_samples)))


def mel_filterbank():
    w = np.zeros((NMELS, NBINS), np.float32)
    lib().so_mel_filterbank(_p(w))
    return w


class Extractor:
    """FeatureExtractor (lib.rs:231-264) restated on the CPU."""

    def __init__(self):
        self._h = lib().so_extractor_create()

    def __del__(self):
        try:
            lib().so_extractor_destroy(self._h)
        except Exception:
            pass

    def extract(self, pcm, fft_mode=0, return_base=False):
        pcm = np.ascontiguousarray(pcm, np.int16)
        n = window_count(pcm.size)
        out = np.zeros((n, FEAT), np.float32)
        base = np.zeros((n, MFCC), np.float32) if return_base else None
        got = lib().so_extract(self._h, _p(pcm), pcm.size, fft_mode, _p(out), _p(base))
        assert got == n
        return (out, base) if return_base else out

    def extract_files(self, pcm, offsets, lengths, fft_mode=1):
        pcm = np.ascontiguousarray(pcm, np.int16)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        lengths = np.ascontiguousarray(lengths, np.uint64)
        nw = np.array([window_count(int(x)) for x in lengths], np.uint64)
        woff = np.zeros(len(nw), np.uint64)
        woff[1:] = np.cumsum(nw)[:-1]
        out = np.zeros((int(nw.sum()), FEAT), np.float32)
        lib().so_extract_files(self._h, _p(pcm), _p(offsets), _p(lengths), len(lengths), fft_mode,
                               _p(out), _p(woff))
        return out, woff, nw


def finish_features(base):
    base = np.ascontiguousarray(base, np.float32)
    out = np.zeros((base.shape[0], FEAT), np.float32)
    lib().so_finish_features(_p(base), base.shape[0], _p(out))
    return out


class Net:
    """SimpleNeuralNet (lib.rs:744-1060) restated on the CPU."""

    NAMES = ("w1", "b1", "w2", "b2", "w3", "b3")

    def __init__(self, inp, h1, h2, out):
        self.inp, self.h1, self.h2, self.out = inp, h1, h2, out
        self._h = lib().so_net_create(inp, h1, h2, out)

    def __del__(self):
        try:
            lib().so_net_destroy(self._h)
        except Exception:
            pass

    def shapes(self):
        return [(self.inp, self.h1), (self.h1,), (self.h1, self.h2), (self.h2,), (self.h2, self.out), (self.out,)]

    def _view(self, i):
        shp = self.shapes()[i]
        n = int(np.prod(shp))
        ptr = lib().so_net_param(self._h, i)
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shp)

    def set_params(self, params):
        for i, p in enumerate(params):
            self._view(i)[...] = np.asarray(p, np.float32).reshape(self.shapes()[i])

    def get_params(self):
        return [self._view(i).copy() for i in range(6)]

    def param_count(self):
        return int(lib().so_net_param_count(self._h))

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.inp)
        out = np.zeros((x.shape[0], self.out), np.float32)
        lib().so_net_forward(self._h, _p(x), x.shape[0], _p(out))
        return out

    def embed(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.inp)
        out = np.zeros((x.shape[0], self.h2), np.float32)
        lib().so_net_embed(self._h, _p(x), x.shape[0], _p(out))
        return out

    def forward_embedding(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.inp)
        out = np.zeros((x.shape[0], self.h2), np.float32)
        lib().so_net_forward_embedding(self._h, _p(x), x.shape[0], _p(out))
        return out

    def grad(self, x, target=None, labels=None):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.inp)
        g = np.zeros(self.param_count(), np.float32)
        t = np.ascontiguousarray(target, np.float32) if target is not None else None
        lab = np.ascontiguousarray(labels, np.int32) if labels is not None else None
        lib().so_net_grad(self._h, _p(x), x.shape[0], _p(t), _p(lab), _p(g))
        return g

    def apply(self, g, scale):
        g = np.ascontiguousarray(g, np.float32)
        lib().so_net_apply(self._h, _p(g), float(scale))

    def train_batch(self, x, target, lr):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.inp)
        t = np.ascontiguousarray(target, np.float32)
        lib().so_net_train_batch(self._h, _p(x), x.shape[0], _p(t), float(lr))

    def train(self, x, target, lr):
        x = np.ascontiguousarray(x, np.float32)
        t = np.ascontiguousarray(target, np.float32)
        lib().so_net_train(self._h, _p(x), _p(t), float(lr))

    def pretrain_from_features(self, windows, target_class, num_classes, epochs, lr, dropout, batch_size,
                               perm=None, keep=None):
        w = np.ascontiguousarray(windows, np.float32).reshape(-1, FEAT)
        pm = np.ascontiguousarray(perm, np.uint32) if perm is not None else None
        kp = np.ascontiguousarray(keep, np.uint8) if keep is not None else None
        return float(lib().so_pretrain_from_features(self._h, _p(w), w.shape[0], target_class, num_classes,
                                                     epochs, float(lr), float(dropout), batch_size,
                                                     _p(pm), _p(kp)))

    def identify_from_features(self, feats, threshold):
        f = np.ascontiguousarray(feats, np.float32).reshape(-1, FEAT)
        ids = np.zeros(max(self.out, 1), np.uint32)
        counts = np.zeros(max(self.out, 1), np.uint64)
        n = lib().so_identify_from_features(self._h, _p(f), f.shape[0], float(threshold), _p(ids), _p(counts))
        return ids[:n].copy(), counts[: self.out].copy()

    def identify_speaker_feats(self, feats):
        """identify_speaker (lib.rs:1285-1303) on feature rows -> (speaker id, per-class sums)."""
        f = np.ascontiguousarray(feats, np.float32).reshape(-1, FEAT)
        sums = np.zeros(max(self.out, 1), np.float32)
        best = lib().so_identify_speaker_feats(self._h, _p(f), f.shape[0], _p(sums))
        return int(best), sums[: self.out].copy()

    def identify_speaker_with_threshold_feats(self, feats, threshold):
        """lib.rs:1346-1377; None when the reference returns None."""
        f = np.ascontiguousarray(feats, np.float32).reshape(-1, FEAT)
        r = lib().so_identify_speaker_with_threshold_feats(self._h, _p(f), f.shape[0], float(threshold))
        return None if r < 0 else int(r)

    def identify_speaker_list(self, ext, pcm, threshold, fft_mode=0):
        pcm = np.ascontiguousarray(pcm, np.int16)
        ids = np.zeros(max(self.out, 1), np.uint32)
        n = lib().so_identify_speaker_list(self._h, ext._h, _p(pcm), pcm.size, float(threshold), fft_mode, _p(ids))
        return ids[:n].copy()


def augment(samples, gain, shift, noise):
    """augment (lib.rs:103-116) with explicit draws; noise[n] are the per-sample values of
    gen_range(-noise_level..noise_level)."""
    x = np.ascontiguousarray(samples, np.int16)
    nz = np.ascontiguousarray(noise, np.float32)
    out = np.empty_like(x)
    lib().so_augment(_p(x), x.size, gain, shift, _p(nz), _p(out))
    return out


def seed(s):
    lib().so_seed(int(s))


# ---- speaker-embedding bookkeeping (host-side arithmetic of the reference, numpy f32) ----------
def _fsum(x):
    """f32 sum in index order, like Rust's iter().sum::<f32>()"""
    x = np.asarray(x, np.float32)
    return np.float32(np.cumsum(x, dtype=np.float32)[-1]) if x.size else np.float32(0.0)


def normalize(v):
    """normalize, lib.rs:132-139: unit L2 norm unless the norm is <= 1e-6"""
    v = np.array(v, np.float32)
    norm = np.sqrt(_fsum(v * v))
    return v / norm if norm > np.float32(1e-6) else v


def average_vectors(vectors):
    """average_vectors, lib.rs:144-159: element-wise mean, then normalize; [] for no vectors"""
    if len(vectors) == 0:
        return np.zeros(0, np.float32)
    acc = np.zeros(len(vectors[0]), np.float32)
    for v in vectors:
        acc = acc + np.asarray(v, np.float32)
    return normalize(acc / np.float32(len(vectors)))


def cosine_similarity(a, b):
    """cosine_similarity, lib.rs:1532-1541 (0.0 when either norm is 0)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    n = min(a.size, b.size)
    dot = _fsum(a[:n] * b[:n])
    na, nb = np.sqrt(_fsum(a * a)), np.sqrt(_fsum(b * b))
    return np.float32(0.0) if na == 0 or nb == 0 else np.float32(dot / (na * nb))


def identify_speaker_from_embedding(emb, speaker_embeddings, threshold):
    """identify_speaker_from_embedding, lib.rs:1499-1529: id of the most similar centroid if its
    similarity exceeds the threshold (x 0.7 while fewer than 20 speakers are known), else None
    (the reference's usize::MAX).  speaker_embeddings: {id: centroid}; ids are visited in
    ascending order (the reference's HashMap order is unspecified; it only matters on exact ties)."""
    best_sim, best_id = np.float32(-3.4028235e38), None
    for sid in sorted(speaker_embeddings):
        sim = cosine_similarity(emb, speaker_embeddings[sid])
        if sim > best_sim:
            best_sim, best_id = sim, sid
    dyn = np.float32(threshold) * np.float32(0.7) if len(speaker_embeddings) < 20 else np.float32(threshold)
    return best_id if best_sim > dyn else None


def extract_embedding_from_features(net, feats):
    """extract_embedding_from_features, lib.rs:1450-1471: mean of forward_embedding over the windows, unit norm"""
    feats = np.asarray(feats, np.float32).reshape(-1, 60)
    e = net.forward_embedding(feats)
    acc = np.zeros(e.shape[1], np.float32)
    for row in e:
        acc = acc + row
    if len(e):
        acc = acc / np.float32(len(e))
    return normalize(acc)


def median_embedding_from_features(net, feats):
    """median_embedding_from_features, lib.rs:1474-1495: per-dimension median (mean of the two middle
    values for an even count) of forward_embedding, unit norm; zeros for no windows"""
    feats = np.asarray(feats, np.float32).reshape(-1, 60)
    if len(feats) == 0:
        return np.zeros(getattr(net, "h2", 0), np.float32)
    e = np.sort(net.forward_embedding(feats), axis=0)
    mid = len(e) // 2
    med = (e[mid - 1] + e[mid]) / np.float32(2.0) if len(e) % 2 == 0 else e[mid]
    return normalize(med.astype(np.float32))


def compute_speaker_embeddings(net, feats_per_speaker, embedding_size):
    """compute_speaker_embeddings, lib.rs:1555-1599, with the cached features of every speaker's
    files handed in (feats_per_speaker[s] = list of (n, 60) arrays): per speaker (mean embedding,
    mean similarity to it, standard deviation of the similarities); zeros for a speaker without files."""
    out = []
    for files in feats_per_speaker:
        embeds = [median_embedding_from_features(net, f) for f in files]
        if not embeds:
            out.append((np.zeros(embedding_size, np.float32), np.float32(0.0), np.float32(0.0)))
            continue
        mean = np.zeros(embedding_size, np.float32)
        for e in embeds:
            mean = mean + e
        mean = normalize(mean / np.float32(len(embeds)))
        sims = np.array([cosine_similarity(e, mean) for e in embeds], np.float32)
        mean_sim = _fsum(sims) / np.float32(len(sims))
        var = _fsum((sims - mean_sim) * (sims - mean_sim)) / np.float32(len(sims))
        out.append((mean, np.float32(mean_sim), np.float32(np.sqrt(var))))
    return out

