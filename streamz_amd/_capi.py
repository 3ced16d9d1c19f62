"""ctypes binding of include/streamz_hip.h.  Names follow the reference's Rust API
(streamz-rs/src/lib.rs): FeatureExtractor, SimpleNeuralNet, pretrain_from_features,
train_from_feature_map, identify_speaker_list, load_cached_features."""
import ctypes as C
import os
import sys

import numpy as np

FEATURE_SIZE = 60

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("STREAMZ_HIP_LIBRARY") or os.path.join(_HERE, "libstreamz_hip.so")   # (the override is for A/B timing of two builds)
_lib = None


class StreamzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"streamz_hip error {code}: {msg}")
        self.code = code


def library_path():
    return _SO


_SIGS = {
    # name: (restype, argtypes)
    "szh_init": (C.c_int, [C.c_int]),
    "szh_device_count": (C.c_int, []),
    "szh_shutdown": (C.c_int, []),
    "szh_last_error": (C.c_char_p, []),
    "szh_free": (None, [C.c_void_p]),
    "szh_window_count": (C.c_uint64, [C.c_uint64]),
    "szh_extractor_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "szh_extractor_destroy": (None, [C.c_void_p]),
    "szh_extractor_mel_filters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "szh_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "szh_extract_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "szh_plan_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "szh_plan_destroy": (None, [C.c_void_p]),
    "szh_plan_total_windows": (C.c_uint64, [C.c_void_p]),
    "szh_plan_window_offsets": (C.c_int, [C.c_void_p, C.c_void_p]),
    "szh_plan_run_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "szh_extract_dev_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "szh_net_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_void_p)]),
    "szh_net_destroy": (None, [C.c_void_p]),
    "szh_net_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "szh_net_dims": (C.c_int, [C.c_void_p, C.c_void_p]),
    "szh_net_output_size": (C.c_uint32, [C.c_void_p]),
    "szh_net_param_count": (C.c_uint64, [C.c_void_p]),
    "szh_net_get_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "szh_net_set_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "szh_net_add_output_class": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "szh_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "szh_net_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "szh_net_forward_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "szh_net_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]),
    "szh_net_train_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_float]),
    "szh_pretrain_from_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.POINTER(C.c_float)]),
    "szh_augment": (C.c_int, [C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "szh_pretrain_network": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "szh_train_from_feature_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                             C.c_float, C.c_float, C.c_uint32, C.c_uint64, C.POINTER(C.c_float)]),
    "szh_identify_speaker_list": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p,
                                            C.c_uint32, C.POINTER(C.c_uint32)]),
    "szh_identify_speaker_list_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_uint32,
                                                C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]),
    "szh_identify_speaker": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_void_p]),
    "szh_identify_speaker_with_threshold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float,
                                                      C.POINTER(C.c_int64)]),
    "szh_identify_speaker_with_threshold_feats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float,
                                                            C.POINTER(C.c_int64), C.c_void_p]),
    "szh_net_pooled_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "szh_extract_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "szh_identify_from_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_uint32,
                                             C.POINTER(C.c_uint32), C.c_void_p]),
    "szh_net_grad_len": (C.c_uint64, [C.c_void_p]),
    "szh_net_grad_dev": (C.c_void_p, [C.c_void_p]),
    "szh_net_train_epochs_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                               C.c_float, C.c_float, C.c_uint64, C.c_void_p]),
    "szh_net_set_gemm": (C.c_int, [C.c_void_p, C.c_int]),
    "szh_net_get_gemm": (C.c_int, [C.c_void_p]),
    "szh_net_zero_grad_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "szh_net_accumulate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_uint64,
                                         C.c_void_p]),
    "szh_net_apply_grad_dev": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "szh_net_forward_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "szh_net_vote_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p]),
    "szh_comm_unique_id": (C.c_int, [C.c_void_p]),
    "szh_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "szh_comm_destroy": (None, [C.c_void_p]),
    "szh_comm_rank": (C.c_int, [C.c_void_p]),
    "szh_comm_world": (C.c_int, [C.c_void_p]),
    "szh_net_allreduce_grad_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "szh_net_broadcast_params_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "szh_net_record_training_file": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p]),
    "szh_net_file_list": (C.c_int64, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "szh_net_set_dataset_specs": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint16]),
    "szh_net_set_embeddings": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "szh_net_get_embeddings": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32)]),
    "szh_net_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "szh_net_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "szh_load_cached_features": (C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "szh_wav_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "szh_wav_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]),
    "szh_shard_ranges": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "szh_profile_enable": (C.c_int, [C.c_int]),
    "szh_profile_report": (C.c_int64, [C.c_void_p, C.c_uint64]),
}


def exported_symbols():
    """Every entry point include/streamz_hip.h declares."""
    return sorted(_SIGS)


def lib():
    """Load libstreamz_hip.so (built in-tree by __graft_entry__.build()); raise if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise StreamzError(-2, f"{_SO} is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        # PyTorch wheels bundle their own HIP/HSA runtime.  A process may hold only one: if
        # torch is importable, load it first so that this library binds to the same
        # libamdhip64.so.7 instead of bringing in a second copy (torch then finds no GPU).
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise StreamzError(rc, (lib().szh_last_error() or b"").decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def window_count(n_samples):
    return int(lib().szh_window_count(int(n_samples)))


class FeatureExtractor:
    """FeatureExtractor (lib.rs:231-264)."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib().szh_extractor_create(C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().szh_extractor_destroy(self._h)
            self._h = None

    def mel_filters(self):
        w = np.zeros((26, 401), np.float32)
        _check(lib().szh_extractor_mel_filters(self._h, _p(w)))
        return w

    def extract(self, samples):
        pcm = np.ascontiguousarray(samples, np.int16)
        cap = window_count(pcm.size)
        out = np.zeros((cap, FEATURE_SIZE), np.float32)
        n = C.c_uint64()
        _check(lib().szh_extract(self._h, _p(pcm), pcm.size, _p(out), cap, C.byref(n)))
        assert n.value == cap
        return out

    def extract_dev_noalloc(self, pcm_ptr, n_samples, feats_ptr, stream=0):
        """One clip at a device address, through the entry point that allocates nothing."""
        _check(lib().szh_extract_dev_noalloc(self._h, C.c_void_p(pcm_ptr), n_samples, C.c_void_p(feats_ptr),
                                             C.c_void_p(stream)))

    def extract_batch(self, clips):
        clips = [np.ascontiguousarray(c, np.int16) for c in clips]
        outs = [np.zeros((window_count(c.size), FEATURE_SIZE), np.float32) for c in clips]
        n = len(clips)
        pp = (C.c_void_p * n)(*[c.ctypes.data for c in clips])
        po = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ln = np.array([c.size for c in clips], np.uint64)
        _check(lib().szh_extract_batch(self._h, pp, _p(ln), n, po))
        return outs


COMM_ID_BYTES = 128


def comm_unique_id():
    """The id rank 0 draws and hands to the other ranks (bytes)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().szh_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """The ranks of a sharded run (szh_comm_*): RCCL communicator on the current device."""

    def __init__(self, comm_id, rank, world):
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id) if comm_id is not None else None
        _check(lib().szh_comm_create(buf, rank, world, C.byref(h)))
        self._h = h
        self.rank, self.world = rank, world

    def __del__(self):
        if getattr(self, "_h", None):
            lib().szh_comm_destroy(self._h)
            self._h = None

    def allreduce_grad(self, net, stream=0):
        _check(lib().szh_net_allreduce_grad_dev(net._h, self._h, C.c_void_p(stream)))

    def broadcast_params(self, net, root=0, stream=0):
        _check(lib().szh_net_broadcast_params_dev(net._h, self._h, root, C.c_void_p(stream)))


class Plan:
    """Device-resident extraction over a fixed file layout (szh_plan_*)."""

    def __init__(self, extractor, offsets, lengths):
        self._ext = extractor
        off = np.ascontiguousarray(offsets, np.uint64)
        ln = np.ascontiguousarray(lengths, np.uint64)
        h = C.c_void_p()
        _check(lib().szh_plan_create(extractor._h, _p(off), _p(ln), len(off), C.byref(h)))
        self._h = h
        self.n_files = len(off)
        self.total_windows = int(lib().szh_plan_total_windows(h))
        wo = np.zeros(self.n_files + 1, np.uint64)
        _check(lib().szh_plan_window_offsets(h, _p(wo)))
        self.window_offsets = wo

    def __del__(self):
        if getattr(self, "_h", None):
            lib().szh_plan_destroy(self._h)
            self._h = None

    def run(self, pcm_ptr, feats_ptr, stream=0):
        _check(lib().szh_plan_run_dev(self._h, C.c_void_p(pcm_ptr), C.c_void_p(feats_ptr), C.c_void_p(stream)))


class SimpleNeuralNet:
    """SimpleNeuralNet (lib.rs:744-1282)."""

    def __init__(self, input=None, hidden1=None, hidden2=None, output=None, init=None, seed=0, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        flat = None
        if init is not None:
            flat = np.concatenate([np.asarray(a, np.float32).ravel() for a in init]) if isinstance(init, (list, tuple)) \
                else np.ascontiguousarray(init, np.float32)
        _check(lib().szh_net_create(input, hidden1, hidden2, output, _p(flat), seed, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().szh_net_destroy(self._h)
            self._h = None

    # -- shape / parameters
    def dims(self):
        d = np.zeros(4, np.uint32)
        _check(lib().szh_net_dims(self._h, _p(d)))
        return tuple(int(v) for v in d)

    def output_size(self):
        return int(lib().szh_net_output_size(self._h))

    def embedding_size(self):
        return self.dims()[2]

    def param_count(self):
        return int(lib().szh_net_param_count(self._h))

    def shapes(self):
        i, h1, h2, o = self.dims()
        return [(i, h1), (h1,), (h1, h2), (h2,), (h2, o), (o,)]

    def get_flat(self):
        flat = np.zeros(self.param_count(), np.float32)
        _check(lib().szh_net_get_params(self._h, _p(flat)))
        return flat

    def get_params(self):
        flat, out, pos = self.get_flat(), [], 0
        for s in self.shapes():
            n = int(np.prod(s))
            out.append(flat[pos:pos + n].reshape(s).copy())
            pos += n
        return out

    def set_params(self, params):
        flat = np.concatenate([np.asarray(a, np.float32).ravel() for a in params])
        assert flat.size == self.param_count()
        _check(lib().szh_net_set_params(self._h, _p(flat)))

    def clone(self):
        h = C.c_void_p()
        _check(lib().szh_net_clone(self._h, C.byref(h)))
        return SimpleNeuralNet(_handle=h)

    def add_output_class(self, new_col=None, seed=0):
        col = np.ascontiguousarray(new_col, np.float32) if new_col is not None else None
        _check(lib().szh_net_add_output_class(self._h, _p(col), seed))

    # -- forward variants
    def _fwd(self, fn, x, width):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.dims()[0])
        out = np.zeros((x.shape[0], width), np.float32)
        _check(fn(self._h, _p(x), x.shape[0], _p(out)))
        return out

    def forward(self, x):
        return self._fwd(lib().szh_net_forward, x, self.output_size())

    def embed(self, x):
        return self._fwd(lib().szh_net_embed, x, self.dims()[2])

    def forward_embedding(self, x):
        return self._fwd(lib().szh_net_forward_embedding, x, self.dims()[2])

    # -- training
    def train(self, x, target, lr):
        x = np.ascontiguousarray(x, np.float32)
        t = np.ascontiguousarray(target, np.float32)
        _check(lib().szh_net_train(self._h, _p(x), _p(t), lr))

    def train_batch(self, batch, target, lr):
        x = np.ascontiguousarray(batch, np.float32).reshape(-1, self.dims()[0])
        t = np.ascontiguousarray(target, np.float32)
        _check(lib().szh_net_train_batch(self._h, _p(x), x.shape[0], _p(t), lr))

    # -- device-resident building blocks
    def grad_len(self):
        return int(lib().szh_net_grad_len(self._h))

    def grad_ptr(self):
        return int(lib().szh_net_grad_dev(self._h))

    def train_epochs_sharded(self, feats, labels, epochs, lr, dropout=0.0, seed=0, comm=None):
        """One optimiser step per epoch over this rank's windows, gradients summed over comm's ranks
        (szh_net_train_epochs_sharded); returns the last epoch's mean loss over all ranks."""
        f = np.ascontiguousarray(feats, np.float32).reshape(-1, FEATURE_SIZE)
        lab = np.ascontiguousarray(labels, np.int32)
        loss = C.c_float()
        _check(lib().szh_net_train_epochs_sharded(self._h, comm._h if comm is not None else None, _p(f), _p(lab), f.shape[0],
                                                  epochs, lr, dropout, seed, C.byref(loss)))
        return loss.value

    def set_gemm(self, mode):
        """"f32" or "bf16x3": arithmetic of the large batched products (SZH_GEMM_*)"""
        _check(lib().szh_net_set_gemm(self._h, {"f32": 0, "bf16x3": 1}[mode]))

    def get_gemm(self):
        return ("f32", "bf16x3")[lib().szh_net_get_gemm(self._h)]

    def zero_grad_dev(self, stream=0):
        _check(lib().szh_net_zero_grad_dev(self._h, C.c_void_p(stream)))

    def accumulate_dev(self, feats_ptr, labels_ptr, n, dropout=0.0, seed=0, stream=0):
        _check(lib().szh_net_accumulate_dev(self._h, C.c_void_p(feats_ptr), C.c_void_p(labels_ptr), n, dropout, seed,
                                            C.c_void_p(stream)))

    def apply_grad_dev(self, lr, stream=0):
        _check(lib().szh_net_apply_grad_dev(self._h, lr, C.c_void_p(stream)))

    def forward_dev(self, feats_ptr, n, probs_ptr, stream=0):
        _check(lib().szh_net_forward_dev(self._h, C.c_void_p(feats_ptr), n, C.c_void_p(probs_ptr), C.c_void_p(stream)))

    def vote_dev(self, feats_ptr, n, threshold, counts_ptr, stream=0):
        _check(lib().szh_net_vote_dev(self._h, C.c_void_p(feats_ptr), n, threshold, C.c_void_p(counts_ptr),
                                      C.c_void_p(stream)))

    # -- metadata / persistence
    def record_training_file(self, speaker, path):
        _check(lib().szh_net_record_training_file(self._h, speaker, path.encode()))

    def file_lists(self):
        out = []
        for s in range(self.output_size()):
            n = lib().szh_net_file_list(self._h, s, None, 0)
            buf = C.create_string_buffer(int(n) + 1)
            lib().szh_net_file_list(self._h, s, buf, int(n) + 1)
            text = buf.value.decode()
            out.append(text.split("\n") if text else [])
        return out

    def set_dataset_specs(self, sample_rate, bits):
        _check(lib().szh_net_set_dataset_specs(self._h, sample_rate, bits))

    def set_embeddings(self, embeds):
        n = len(embeds)
        dim = len(embeds[0][0]) if n else 0
        v = np.array([e[0] for e in embeds], np.float32).reshape(n, dim)
        m = np.array([e[1] for e in embeds], np.float32)
        s = np.array([e[2] for e in embeds], np.float32)
        _check(lib().szh_net_set_embeddings(self._h, _p(v), _p(m), _p(s), n, dim))

    def embeddings(self):
        n, dim = C.c_uint32(), C.c_uint32()
        _check(lib().szh_net_get_embeddings(self._h, None, None, None, C.byref(n), C.byref(dim)))
        v = np.zeros((n.value, dim.value), np.float32)
        m = np.zeros(n.value, np.float32)
        s = np.zeros(n.value, np.float32)
        _check(lib().szh_net_get_embeddings(self._h, _p(v), _p(m), _p(s), C.byref(n), C.byref(dim)))
        return [(v[i], float(m[i]), float(s[i])) for i in range(n.value)]

    def save(self, path):
        _check(lib().szh_net_save(self._h, os.fsencode(path)))

    @staticmethod
    def load(path):
        h = C.c_void_p()
        _check(lib().szh_net_load(os.fsencode(path), C.byref(h)))
        return SimpleNeuralNet(_handle=h)


class _Rng(C.Structure):
    _fields_ = [("perm", C.c_void_p), ("keep", C.c_void_p), ("seed", C.c_uint64)]


def pretrain_from_features(net, windows, target_class, num_classes, epochs, lr, dropout, batch_size,
                           perm=None, keep=None, seed=0):
    """pretrain_from_features (lib.rs:582-628); perm/keep inject the shuffle and dropout streams."""
    w = np.ascontiguousarray(windows, np.float32).reshape(-1, FEATURE_SIZE)
    pm = np.ascontiguousarray(perm, np.uint32) if perm is not None else None
    kp = np.ascontiguousarray(keep, np.uint8) if keep is not None else None
    rng = _Rng(pm.ctypes.data if pm is not None else None, kp.ctypes.data if kp is not None else None, seed)
    loss = C.c_float()
    _check(lib().szh_pretrain_from_features(net._h, _p(w), w.shape[0], target_class, num_classes, epochs, lr, dropout,
                                            batch_size, C.byref(rng), C.byref(loss)))
    return float(loss.value)


class _AugDraws(C.Structure):
    _fields_ = [("gain", C.c_void_p), ("noise_level", C.c_void_p), ("shift", C.c_void_p), ("noise", C.c_void_p)]


def augment(samples, gain, noise_level, shift, noise=None, seed=0):
    """augment (lib.rs:103-116) with explicit draws; noise[n] in (-noise_level, noise_level) or None."""
    x = np.ascontiguousarray(samples, np.int16)
    nz = np.ascontiguousarray(noise, np.float32) if noise is not None else None
    out = np.empty_like(x)
    _check(lib().szh_augment(_p(x), x.size, gain, noise_level, shift, _p(nz) if nz is not None else None, seed, _p(out)))
    return out


def pretrain_network(net, samples, target_class, num_classes, epochs, lr, dropout, batch_size, extractor,
                     gain=None, noise_level=None, shift=None, noise=None, perm=None, keep=None, seed=0):
    """pretrain_network (lib.rs:348-397); every random draw can be injected."""
    x = np.ascontiguousarray(samples, np.int16)
    arrs = [np.ascontiguousarray(a, t) if a is not None else None
            for a, t in ((gain, np.float32), (noise_level, np.float32), (shift, np.uint64), (noise, np.float32))]
    aug = _AugDraws(*[a.ctypes.data if a is not None else None for a in arrs])
    pm = np.ascontiguousarray(perm, np.uint32) if perm is not None else None
    kp = np.ascontiguousarray(keep, np.uint8) if keep is not None else None
    rng = _Rng(pm.ctypes.data if pm is not None else None, kp.ctypes.data if kp is not None else None, seed)
    loss = C.c_float()
    _check(lib().szh_pretrain_network(net._h, extractor._h, _p(x), x.size, target_class, num_classes, epochs, lr, dropout,
                                      batch_size, C.byref(aug), C.byref(rng), C.byref(loss)))
    return float(loss.value)


def train_from_feature_map(net, feature_map, files, epochs, lr, dropout, batch_size, seed=0):
    """train_from_feature_map (lib.rs:632-665): files is a list of (path, class)."""
    arrs = [np.ascontiguousarray(feature_map[p], np.float32).reshape(-1, FEATURE_SIZE) if p in feature_map else None
            for p, _ in files]
    n = len(files)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data if a is not None else None for a in arrs])
    nw = np.array([a.shape[0] if a is not None else 0 for a in arrs], np.uint64)
    cls = np.array([c for _, c in files], np.uint32)
    loss = C.c_float()
    _check(lib().szh_train_from_feature_map(net._h, ptrs, _p(nw), _p(cls), n, epochs, lr, dropout, batch_size, seed,
                                            C.byref(loss)))
    for (p, c), a in zip(files, arrs):
        if a is not None:
            net.record_training_file(c, p)  # lib.rs:655
    return float(loss.value)


def identify_speaker_list(net, sample, threshold, extractor):
    """identify_speaker_list (lib.rs:1383-1411)."""
    pcm = np.ascontiguousarray(sample, np.int16)
    cap = max(net.output_size(), 1)
    ids = np.zeros(cap, np.uint32)
    n = C.c_uint32()
    _check(lib().szh_identify_speaker_list(net._h, extractor._h, _p(pcm), pcm.size, threshold, _p(ids), cap,
                                           C.byref(n)))
    return [int(v) for v in ids[:n.value]]


def identify_speaker(net, sample, extractor, return_sums=False):
    """identify_speaker (lib.rs:1285-1303)."""
    pcm = np.ascontiguousarray(sample, np.int16)
    sums = np.zeros(max(net.output_size(), 1), np.float32)
    best = C.c_uint32()
    _check(lib().szh_identify_speaker(net._h, extractor._h, _p(pcm), pcm.size, C.byref(best), _p(sums)))
    return (best.value, sums[:net.output_size()].copy()) if return_sums else best.value


def identify_speaker_with_threshold(net, sample, threshold, extractor):
    """identify_speaker_with_threshold (lib.rs:1307-1343); None as the reference's None."""
    pcm = np.ascontiguousarray(sample, np.int16)
    r = C.c_int64()
    _check(lib().szh_identify_speaker_with_threshold(net._h, extractor._h, _p(pcm), pcm.size, threshold, C.byref(r)))
    return None if r.value < 0 else int(r.value)


def identify_speaker_with_threshold_feats(net, windows, threshold, return_sums=False):
    """identify_speaker_with_threshold_feats (lib.rs:1346-1377)."""
    f = np.ascontiguousarray(windows, np.float32).reshape(-1, FEATURE_SIZE)
    sums = np.zeros(max(net.output_size(), 1), np.float32)
    r = C.c_int64()
    _check(lib().szh_identify_speaker_with_threshold_feats(net._h, _p(f), f.shape[0], threshold, C.byref(r), _p(sums)))
    got = None if r.value < 0 else int(r.value)
    return (got, sums[:net.output_size()].copy()) if return_sums else got


def _pooled(net, feats, mode):
    f = np.ascontiguousarray(feats, np.float32).reshape(-1, FEATURE_SIZE)
    out = np.zeros(net.dims()[2], np.float32)
    _check(lib().szh_net_pooled_embedding(net._h, _p(f), f.shape[0], mode, _p(out)))
    return out


def extract_embedding_from_features(net, feats):
    """extract_embedding_from_features (lib.rs:1450-1471)."""
    return _pooled(net, feats, 0)


def median_embedding_from_features(net, feats):
    """median_embedding_from_features (lib.rs:1474-1495)."""
    return _pooled(net, feats, 1)


def extract_embedding(net, sample, extractor):
    """extract_embedding (lib.rs:1418-1447)."""
    pcm = np.ascontiguousarray(sample, np.int16)
    out = np.zeros(net.dims()[2], np.float32)
    _check(lib().szh_extract_embedding(net._h, extractor._h, _p(pcm), pcm.size, _p(out)))
    return out


def identify_speaker_list_dev(net, pcm_dev_ptr, n_samples, threshold, extractor, tile_windows=0):
    """identify_speaker_list for int16 PCM at a device address; returns (ids, vote counts)."""
    cap = max(net.output_size(), 1)
    ids = np.zeros(cap, np.uint32)
    counts = np.zeros(cap, np.uint64)
    n = C.c_uint32()
    _check(lib().szh_identify_speaker_list_dev(net._h, extractor._h, pcm_dev_ptr, n_samples, threshold, tile_windows,
                                               _p(ids), cap, C.byref(n), _p(counts)))
    return [int(v) for v in ids[:n.value]], counts[:net.output_size()].copy()


def identify_from_features(net, feats, threshold):
    f = np.ascontiguousarray(feats, np.float32).reshape(-1, FEATURE_SIZE)
    cap = max(net.output_size(), 1)
    ids = np.zeros(cap, np.uint32)
    counts = np.zeros(cap, np.uint64)
    n = C.c_uint32()
    _check(lib().szh_identify_from_features(net._h, _p(f), f.shape[0], threshold, _p(ids), cap, C.byref(n), _p(counts)))
    return [int(v) for v in ids[:n.value]], counts[:net.output_size()].copy()


def load_cached_features(path, extractor):
    """load_cached_features (lib.rs:558-579)."""
    ptr, n = C.c_void_p(), C.c_uint64()
    _check(lib().szh_load_cached_features(os.fsencode(path), extractor._h, C.byref(ptr), C.byref(n)))
    try:
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n.value * FEATURE_SIZE,)).copy() \
            if n.value else np.zeros(0, np.float32)
    finally:
        lib().szh_free(ptr)
    return arr.reshape(n.value, FEATURE_SIZE)


def wav_read(path):
    ptr, n = C.c_void_p(), C.c_uint64()
    _check(lib().szh_wav_read(os.fsencode(path), C.byref(ptr), C.byref(n)))
    try:
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16)), shape=(n.value,)).copy() if n.value \
            else np.zeros(0, np.int16)
    finally:
        lib().szh_free(ptr)
    return arr


def wav_info(path):
    """(mono sample frames, sample rate, channels) from the file's header alone"""
    n, sr, ch = C.c_uint64(), C.c_uint32(), C.c_uint16()
    _check(lib().szh_wav_info(os.fsencode(path), C.byref(n), C.byref(sr), C.byref(ch)))
    return int(n.value), int(sr.value), int(ch.value)


def shard_ranges(n_windows, world):
    """szh_shard_ranges: world + 1 file indices; rank r takes files [out[r], out[r + 1])"""
    nw = np.ascontiguousarray(n_windows, np.uint64)
    out = np.zeros(world + 1, np.uint32)
    _check(lib().szh_shard_ranges(_p(nw), nw.size, world, _p(out)))
    return [int(v) for v in out]


def profile_enable(on=True):
    _check(lib().szh_profile_enable(1 if on else 0))


def profile_report():
    """{tag: (launches, total_ms)} of the kernels launched since the previous report."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().szh_profile_report(buf, len(buf))
    if n < 0:
        _check(int(n))
    out = {}
    for line in buf.value.decode().splitlines():
        tag, cnt, ms = line.split()
        out[tag] = (int(cnt), float(ms))
    return out
