// 16-bit PCM WAV ingest (reference lib.rs:509-529 wav arm, lib.rs:172-183 downmix).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace sz {

struct WavInfo {
    uint32_t sample_rate = 0;
    uint16_t channels = 0;
    uint16_t bits = 0;
};

// Interleaved samples as stored.  false + err when the file is missing or is not RIFF/WAVE
// integer PCM with 16 bits per sample.
bool wav_read_raw(const std::string& path, std::vector<int16_t>& samples, WavInfo& info, std::string& err);
// Header walk only (no sample is read): format and the number of sample frames of the data chunk, as
// wav_read_raw followed by downmix_to_mono would deliver them.
bool wav_probe(const std::string& path, WavInfo& info, uint64_t& n_frames, std::string& err);
// downmix_to_mono, lib.rs:172-183: per-frame integer mean (i32 sum / channels, truncating).
std::vector<int16_t> downmix_to_mono(const std::vector<int16_t>& samples, size_t channels);
// Canonical 44-byte-header mono/multi-channel 16-bit writer (used by tests and the bench).
bool wav_write(const std::string& path, const int16_t* samples, size_t n, uint32_t sample_rate, uint16_t channels,
               std::string& err);

}  // namespace sz
