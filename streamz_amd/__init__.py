"""streamz_amd: ctypes access to libstreamz_hip.so, the MI355X implementation of the StreamZ
MFCC + SimpleNeuralNet path.

The compute lives in the shared library (HIP kernels behind a C ABI, include/streamz_hip.h).
This package only loads it and mirrors the reference's public names for tests and the
benchmark.  There is no CPU fallback: importing works without a GPU (so that symbols can be
checked), but every compute call raises StreamzError when no HIP device is usable, and
loading raises if the library has not been built.
"""
from . import _capi
from ._capi import (Comm, comm_unique_id, extract_embedding, extract_embedding_from_features, median_embedding_from_features,
                    FEATURE_SIZE, FeatureExtractor, augment, pretrain_network, Plan, SimpleNeuralNet, StreamzError, exported_symbols,
                    identify_from_features, identify_speaker, identify_speaker_with_threshold,
                    identify_speaker_with_threshold_feats, identify_speaker_list, identify_speaker_list_dev,
                    lib, library_path, load_cached_features, pretrain_from_features, profile_enable,
                    profile_report, shard_ranges, train_from_feature_map, wav_info, wav_read, window_count)

__all__ = ["Comm", "comm_unique_id", "extract_embedding", "extract_embedding_from_features", "median_embedding_from_features", "FEATURE_SIZE", "FeatureExtractor", "augment", "pretrain_network", "Plan", "SimpleNeuralNet", "StreamzError", "exported_symbols",
           "identify_from_features", "identify_speaker", "identify_speaker_with_threshold",
           "identify_speaker_with_threshold_feats", "identify_speaker_list", "identify_speaker_list_dev",
           "lib", "library_path", "load_cached_features", "pretrain_from_features", "profile_enable",
           "profile_report", "shard_ranges", "train_from_feature_map", "wav_info", "wav_read", "window_count"]
