#include "wav.h"

#include <cstdio>
#include <cstring>

namespace sz {

namespace {
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
}  // namespace

bool wav_read_raw(const std::string& path, std::vector<int16_t>& samples, WavInfo& info, std::string& err)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(n > 0 ? (size_t)n : 0);
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size() || buf.size() < 12 || std::memcmp(buf.data(), "RIFF", 4) != 0 ||
        std::memcmp(buf.data() + 8, "WAVE", 4) != 0) {
        err = path + ": not a RIFF/WAVE file";
        return false;
    }
    size_t pos = 12;
    bool have_fmt = false;
    uint16_t tag = 0;
    while (pos + 8 <= buf.size()) {
        const uint32_t clen = rd32(buf.data() + pos + 4);
        const uint8_t* body = buf.data() + pos + 8;
        const size_t avail = buf.size() - (pos + 8);
        if (std::memcmp(buf.data() + pos, "fmt ", 4) == 0) {
            if (clen < 16 || avail < 16) { err = path + ": short fmt chunk"; return false; }
            tag = rd16(body);
            info.channels = rd16(body + 2);
            info.sample_rate = rd32(body + 4);
            info.bits = rd16(body + 14);
            if (tag == 0xFFFE && clen >= 26 && avail >= 26) tag = rd16(body + 24);  // extensible: sub-format
            have_fmt = true;
        } else if (std::memcmp(buf.data() + pos, "data", 4) == 0) {
            if (!have_fmt) { err = path + ": data chunk before fmt chunk"; return false; }
            if (tag != 1 || info.bits != 16 || info.channels == 0) {
                err = path + ": only 16-bit integer PCM is supported";
                return false;
            }
            const size_t bytes = clen <= avail ? clen : avail;
            samples.resize(bytes / 2);
            if (!samples.empty()) std::memcpy(samples.data(), body, samples.size() * 2);
            return true;
        }
        pos += 8 + (size_t)clen + (clen & 1);
    }
    err = path + ": no data chunk";
    return false;
}

bool wav_probe(const std::string& path, WavInfo& info, uint64_t& n_frames, std::string& err)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    uint8_t h[40];
    if (size < 12 || std::fread(h, 1, 12, f) != 12 || std::memcmp(h, "RIFF", 4) != 0 || std::memcmp(h + 8, "WAVE", 4) != 0) {
        err = path + ": not a RIFF/WAVE file";
        return false;
    }
    long pos = 12;
    bool have_fmt = false;
    uint16_t tag = 0;
    while (pos + 8 <= size) {
        if (std::fseek(f, pos, SEEK_SET) != 0 || std::fread(h, 1, 8, f) != 8) break;
        const uint32_t clen = rd32(h + 4);
        const uint64_t avail = (uint64_t)(size - (pos + 8));
        if (std::memcmp(h, "fmt ", 4) == 0) {
            const size_t want = clen >= 26 && avail >= 26 ? 26 : 16;
            if (clen < 16 || avail < 16 || std::fread(h, 1, want, f) != want) { err = path + ": short fmt chunk"; return false; }
            tag = rd16(h);
            info.channels = rd16(h + 2);
            info.sample_rate = rd32(h + 4);
            info.bits = rd16(h + 14);
            if (tag == 0xFFFE && want == 26) tag = rd16(h + 24);
            have_fmt = true;
        } else if (std::memcmp(h, "data", 4) == 0) {
            if (!have_fmt) { err = path + ": data chunk before fmt chunk"; return false; }
            if (tag != 1 || info.bits != 16 || info.channels == 0) { err = path + ": only 16-bit integer PCM is supported"; return false; }
            const uint64_t samples = ((uint64_t)clen <= avail ? (uint64_t)clen : avail) / 2;
            n_frames = info.channels <= 1 ? samples : (samples + info.channels - 1) / info.channels;  // downmix keeps a ragged last frame
            return true;
        }
        pos += 8 + (long)clen + (long)(clen & 1);
    }
    err = path + ": no data chunk";
    return false;
}

std::vector<int16_t> downmix_to_mono(const std::vector<int16_t>& samples, size_t channels)
{
    if (channels <= 1) return samples;
    std::vector<int16_t> out;
    out.reserve(samples.size() / channels + 1);
    for (size_t i = 0; i < samples.size(); i += channels) {
        int32_t sum = 0;
        for (size_t c = 0; c < channels && i + c < samples.size(); ++c) sum += samples[i + c];
        out.push_back((int16_t)(sum / (int32_t)channels));
    }
    return out;
}

bool wav_write(const std::string& path, const int16_t* samples, size_t n, uint32_t sample_rate, uint16_t channels,
               std::string& err)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    const uint32_t data_bytes = (uint32_t)(n * 2);
    uint8_t h[44];
    auto w32 = [&](int o, uint32_t v) { for (int i = 0; i < 4; ++i) h[o + i] = (v >> (8 * i)) & 0xff; };
    auto w16 = [&](int o, uint16_t v) { h[o] = v & 0xff; h[o + 1] = v >> 8; };
    std::memcpy(h, "RIFF", 4); w32(4, 36 + data_bytes); std::memcpy(h + 8, "WAVEfmt ", 8); w32(16, 16);
    w16(20, 1); w16(22, channels); w32(24, sample_rate); w32(28, sample_rate * channels * 2);
    w16(32, (uint16_t)(channels * 2)); w16(34, 16); std::memcpy(h + 36, "data", 4); w32(40, data_bytes);
    bool ok = std::fwrite(h, 1, 44, f) == 44;
    if (ok && n) ok = std::fwrite(samples, 2, n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) err = "short write on " + path;
    return ok;
}

}  // namespace sz
