// C++ host-side mirror of the reference's public Rust API for the feature-extraction and
// SimpleNeuralNet path (crate streamz_rs, streamz-rs/src/lib.rs), written over the C ABI of
// libstreamz_hip.so.  Names, argument meaning and error behaviour follow the reference:
// numeric functions are infallible and return empty / 0.0 on empty input, I/O functions throw
// streamz::Error (the reference returns Err).  A failure of the device itself also throws.
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "streamz_hip.h"

namespace streamz {

constexpr uint32_t DEFAULT_SAMPLE_RATE = SZH_SAMPLE_RATE;  // lib.rs:25
constexpr size_t WINDOW_SIZE = SZH_WINDOW_SIZE;            // lib.rs:26
constexpr size_t MFCC_SIZE = SZH_MFCC_SIZE;                // lib.rs:28
constexpr size_t FEATURE_SIZE = SZH_FEATURE_SIZE;          // lib.rs:30-34
constexpr float DEFAULT_DROPOUT = 0.2f;                    // lib.rs:36

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc)
{
    if (rc != SZH_OK) throw Error(rc, szh_last_error());
}

// Row-major [n][60] feature windows of one file (the reference's Vec<Vec<f32>>).
struct Features {
    std::vector<float> data;
    size_t len() const { return data.size() / FEATURE_SIZE; }
    bool is_empty() const { return data.empty(); }
    const float* row(size_t i) const { return data.data() + i * FEATURE_SIZE; }
};

// normalize, lib.rs:132-139
inline void normalize(std::vector<float>& v)
{
    float s = 0.0f;
    for (float x : v) s += x * x;
    const float norm = std::sqrt(s);
    if (norm > 1e-6f)
        for (float& x : v) x /= norm;
}
// cosine_similarity, lib.rs:1532-1541
inline float cosine_similarity(const std::vector<float>& a, const std::vector<float>& b)
{
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    const size_t n = std::min(a.size(), b.size());
    for (size_t i = 0; i < n; ++i) dot += a[i] * b[i];
    for (float x : a) na += x * x;
    for (float x : b) nb += x * x;
    na = std::sqrt(na);
    nb = std::sqrt(nb);
    return (na == 0.0f || nb == 0.0f) ? 0.0f : dot / (na * nb);
}
// average_vectors, lib.rs:144-159
inline std::vector<float> average_vectors(const std::vector<std::vector<float>>& vs)
{
    if (vs.empty()) return {};
    std::vector<float> avg(vs[0].size(), 0.0f);
    for (const auto& v : vs)
        for (size_t i = 0; i < v.size() && i < avg.size(); ++i) avg[i] += v[i];
    for (float& a : avg) a /= (float)vs.size();
    normalize(avg);
    return avg;
}

// FeatureExtractor, lib.rs:231-264
class FeatureExtractor {
public:
    FeatureExtractor() { check(szh_extractor_create(&h_)); }
    ~FeatureExtractor() { szh_extractor_destroy(h_); }
    FeatureExtractor(const FeatureExtractor&) = delete;
    FeatureExtractor& operator=(const FeatureExtractor&) = delete;

    Features extract(const std::vector<int16_t>& samples) const
    {
        Features f;
        const uint64_t n = szh_window_count(samples.size());
        f.data.resize(n * FEATURE_SIZE);
        uint64_t got = 0;
        check(szh_extract(h_, samples.data(), samples.size(), f.data.data(), n, &got));
        return f;
    }
    // the file-parallel loop of main.rs:500-508
    std::vector<Features> extract_batch(const std::vector<std::vector<int16_t>>& clips) const
    {
        std::vector<Features> out(clips.size());
        std::vector<const int16_t*> p(clips.size());
        std::vector<uint64_t> n(clips.size());
        std::vector<float*> o(clips.size());
        for (size_t i = 0; i < clips.size(); ++i) {
            p[i] = clips[i].data();
            n[i] = clips[i].size();
            out[i].data.resize(szh_window_count(n[i]) * FEATURE_SIZE);
            o[i] = out[i].data.data();
        }
        check(szh_extract_batch(h_, p.data(), n.data(), (uint32_t)clips.size(), o.data()));
        return out;
    }
    szh_extractor* handle() const { return h_; }

private:
    szh_extractor* h_ = nullptr;
};

// SimpleNeuralNet, lib.rs:744-1282
class SimpleNeuralNet {
public:
    SimpleNeuralNet(size_t input, size_t hidden1, size_t hidden2, size_t output, uint64_t seed = 0x5EED)
    {
        check(szh_net_create((uint32_t)input, (uint32_t)hidden1, (uint32_t)hidden2, (uint32_t)output, nullptr, seed, &h_));
    }
    explicit SimpleNeuralNet(szh_net* adopt) : h_(adopt) {}
    SimpleNeuralNet(const SimpleNeuralNet& o) { check(szh_net_clone(o.h_, &h_)); }  // #[derive(Clone)]
    SimpleNeuralNet(SimpleNeuralNet&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    SimpleNeuralNet& operator=(SimpleNeuralNet&& o) noexcept { std::swap(h_, o.h_); return *this; }
    SimpleNeuralNet& operator=(const SimpleNeuralNet&) = delete;
    ~SimpleNeuralNet() { szh_net_destroy(h_); }

    size_t output_size() const { return szh_net_output_size(h_); }
    size_t embedding_size() const { uint32_t d[4]; check(szh_net_dims(h_, d)); return d[2]; }
    void add_output_class(uint64_t seed = 0xC1A55) { check(szh_net_add_output_class(h_, nullptr, seed)); }
    void set_dataset_specs(uint32_t sample_rate, uint16_t bits) { check(szh_net_set_dataset_specs(h_, sample_rate, bits)); }
    void record_training_file(size_t cls, const std::string& path) { check(szh_net_record_training_file(h_, (uint32_t)cls, path.c_str())); }
    std::vector<std::vector<std::string>> file_lists() const
    {
        std::vector<std::vector<std::string>> out(output_size());
        for (size_t s = 0; s < out.size(); ++s) {
            const int64_t n = szh_net_file_list(h_, (uint32_t)s, nullptr, 0);
            if (n <= 0) continue;
            std::string buf((size_t)n + 1, '\0');
            szh_net_file_list(h_, (uint32_t)s, &buf[0], buf.size());
            buf.resize((size_t)n);
            size_t pos = 0;
            while (pos <= buf.size()) {
                size_t nl = buf.find('\n', pos);
                if (nl == std::string::npos) nl = buf.size();
                out[s].push_back(buf.substr(pos, nl - pos));
                pos = nl + 1;
            }
        }
        return out;
    }

    std::vector<float> forward(const float* bits) const
    {
        std::vector<float> o(output_size());
        uint32_t d[4];
        check(szh_net_dims(h_, d));
        (void)d;
        check(szh_net_forward(h_, bits, 1, o.data()));
        return o;
    }
    // all windows at once: [n][output_size]
    std::vector<float> forward_rows(const float* x, size_t n) const
    {
        std::vector<float> o(n * output_size());
        check(szh_net_forward(h_, x, n, o.data()));
        return o;
    }
    std::vector<float> embed_rows(const float* x, size_t n) const
    {
        std::vector<float> o(n * embedding_size());
        check(szh_net_embed(h_, x, n, o.data()));
        return o;
    }
    std::vector<float> forward_embedding_rows(const float* x, size_t n) const
    {
        std::vector<float> o(n * embedding_size());
        check(szh_net_forward_embedding(h_, x, n, o.data()));
        return o;
    }
    void train(const float* bits, const float* target, float lr) { check(szh_net_train(h_, bits, target, lr)); }
    void train_batch(const float* batch, size_t rows, const float* target, float lr) { check(szh_net_train_batch(h_, batch, rows, target, lr)); }

    using Embedding = std::pair<std::vector<float>, std::pair<float, float>>;  // (mean, (mean_sim, std_sim))
    void set_embeddings(const std::vector<Embedding>& e)
    {
        const uint32_t dim = e.empty() ? 0 : (uint32_t)e[0].first.size();
        std::vector<float> v, m, s;
        for (const auto& x : e) {
            v.insert(v.end(), x.first.begin(), x.first.end());
            m.push_back(x.second.first);
            s.push_back(x.second.second);
        }
        check(szh_net_set_embeddings(h_, v.data(), m.data(), s.data(), (uint32_t)e.size(), dim));
    }
    std::vector<Embedding> embeddings() const
    {
        uint32_t n = 0, dim = 0;
        check(szh_net_get_embeddings(h_, nullptr, nullptr, nullptr, &n, &dim));
        std::vector<float> v((size_t)n * dim), m(n), s(n);
        check(szh_net_get_embeddings(h_, v.data(), m.data(), s.data(), &n, &dim));
        std::vector<Embedding> out;
        for (uint32_t i = 0; i < n; ++i)
            out.push_back({std::vector<float>(v.begin() + (size_t)i * dim, v.begin() + (size_t)(i + 1) * dim), {m[i], s[i]}});
        return out;
    }

    void save(const std::string& path) const { check(szh_net_save(h_, path.c_str())); }
    static SimpleNeuralNet load(const std::string& path)
    {
        szh_net* h = nullptr;
        check(szh_net_load(path.c_str(), &h));
        return SimpleNeuralNet(h);
    }
    szh_net* handle() const { return h_; }

private:
    szh_net* h_ = nullptr;
};

// load_and_resample_file, lib.rs:509-529 (WAV at 44.1 kHz only; anything else throws)
inline std::vector<int16_t> load_and_resample_file(const std::string& path)
{
    int16_t* p = nullptr;
    uint64_t n = 0;
    check(szh_wav_read(path.c_str(), &p, &n));
    std::vector<int16_t> out(p, p + n);
    szh_free(p);
    return out;
}
// batch_resample, lib.rs:541-547: unreadable files are silently dropped
inline std::vector<std::pair<std::string, std::vector<int16_t>>> batch_resample(const std::vector<std::string>& paths)
{
    std::vector<std::pair<std::string, std::vector<int16_t>>> out;
    for (const auto& p : paths) {
        try { out.emplace_back(p, load_and_resample_file(p)); } catch (const Error&) {}
    }
    return out;
}

// load_cached_features, lib.rs:558-579
inline Features load_cached_features(const std::string& path, const FeatureExtractor& ext)
{
    float* p = nullptr;
    uint64_t n = 0;
    check(szh_load_cached_features(path.c_str(), ext.handle(), &p, &n));
    Features f;
    f.data.assign(p, p + n * FEATURE_SIZE);
    szh_free(p);
    return f;
}

// pretrain_from_features, lib.rs:582-628
inline float pretrain_from_features(SimpleNeuralNet& net, const Features& windows, size_t target_class, size_t num_classes,
                                    size_t epochs, float lr, float dropout, size_t batch_size, uint64_t seed = 0)
{
    float loss = 0.0f;
    szh_rng rng{nullptr, nullptr, seed};
    check(szh_pretrain_from_features(net.handle(), windows.data.data(), windows.len(), (uint32_t)target_class,
                                     (uint32_t)num_classes, (uint32_t)epochs, lr, dropout, (uint32_t)batch_size, &rng, &loss));
    return loss;
}

// augment, lib.rs:103-116 (gain, additive noise, circular shift); the draws come from `seed`
inline std::vector<int16_t> augment(const std::vector<int16_t>& samples, uint64_t seed = 0)
{
    std::vector<int16_t> out(samples.size());
    if (samples.empty()) return out;
    // the same three draws the reference makes per call (lib.rs:105-107), from a counter-based generator
    auto u01 = [&](uint64_t k) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (k + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (float)(z >> 40) * (1.0f / 16777216.0f);
    };
    const float level = 0.005f * u01(0), gain = 0.95f + 0.1f * u01(1);
    const uint64_t range = std::min<uint64_t>(samples.size(), 800);
    const uint64_t shift = (uint64_t)(u01(2) * (float)range) % range;
    check(szh_augment(samples.data(), samples.size(), gain, level, shift, nullptr, seed, out.data()));
    return out;
}

// pretrain_network, lib.rs:348-397
inline float pretrain_network(SimpleNeuralNet& net, const std::vector<int16_t>& samples, size_t target_class,
                              size_t num_classes, size_t epochs, float lr, float dropout, size_t batch_size,
                              const FeatureExtractor& extractor, uint64_t seed = 0)
{
    float loss = 0.0f;
    szh_rng rng{nullptr, nullptr, seed};
    check(szh_pretrain_network(net.handle(), extractor.handle(), samples.data(), samples.size(), (uint32_t)target_class,
                               (uint32_t)num_classes, (uint32_t)epochs, lr, dropout, (uint32_t)batch_size, nullptr, &rng, &loss));
    return loss;
}

// train_from_feature_map, lib.rs:632-665
inline float train_from_feature_map(SimpleNeuralNet& net, const std::map<std::string, Features>& feature_map,
                                    const std::vector<std::pair<std::string, size_t>>& files, size_t epochs, float lr,
                                    float dropout, size_t batch_size, uint64_t seed = 0)
{
    float total = 0.0f;
    size_t count = 0;
    for (const auto& fc : files) {
        auto it = feature_map.find(fc.first);
        if (it == feature_map.end()) continue;
        const float loss = pretrain_from_features(net, it->second, fc.second, net.output_size(), epochs, lr, dropout,
                                                  batch_size, seed + count);
        net.record_training_file(fc.second, fc.first);
        total += loss;
        count += 1;
    }
    return count > 0 ? total / (float)count : 0.0f;
}

// identify_speaker_list, lib.rs:1383-1411
inline std::vector<size_t> identify_speaker_list(const SimpleNeuralNet& net, const std::vector<int16_t>& sample,
                                                 float threshold, const FeatureExtractor& ext)
{
    std::vector<uint32_t> ids(std::max<size_t>(net.output_size(), 1));
    uint32_t n = 0;
    check(szh_identify_speaker_list(net.handle(), ext.handle(), sample.data(), sample.size(), threshold, ids.data(),
                                    (uint32_t)ids.size(), &n));
    return std::vector<size_t>(ids.begin(), ids.begin() + n);
}

// identify_speaker, lib.rs:1285-1303
inline size_t identify_speaker(const SimpleNeuralNet& net, const std::vector<int16_t>& sample, const FeatureExtractor& ext)
{
    uint32_t id = 0;
    check(szh_identify_speaker(net.handle(), ext.handle(), sample.data(), sample.size(), &id, nullptr));
    return id;
}

// identify_speaker_with_threshold, lib.rs:1307-1343 (SIZE_MAX = None)
inline size_t identify_speaker_with_threshold(const SimpleNeuralNet& net, const std::vector<int16_t>& sample, float threshold,
                                              const FeatureExtractor& ext)
{
    int64_t id = -1;
    check(szh_identify_speaker_with_threshold(net.handle(), ext.handle(), sample.data(), sample.size(), threshold, &id));
    return id < 0 ? (size_t)-1 : (size_t)id;
}

// identify_speaker_with_threshold_feats, lib.rs:1346-1377 (SIZE_MAX = None)
inline size_t identify_speaker_with_threshold_feats(const SimpleNeuralNet& net, const Features& windows, float threshold)
{
    int64_t id = -1;
    check(szh_identify_speaker_with_threshold_feats(net.handle(), windows.data.data(), windows.len(), threshold, &id, nullptr));
    return id < 0 ? (size_t)-1 : (size_t)id;
}

// extract_embedding, lib.rs:1418-1447: median of embed (tanh layer) over the sample's windows
inline std::vector<float> extract_embedding(const SimpleNeuralNet& net, const std::vector<int16_t>& sample,
                                            const FeatureExtractor& ext)
{
    std::vector<float> emb(net.embedding_size(), 0.0f);
    check(szh_extract_embedding(net.handle(), ext.handle(), sample.data(), sample.size(), emb.data()));
    return emb;
}

// extract_embedding_from_features, lib.rs:1450-1471: mean of forward_embedding, unit norm
inline std::vector<float> extract_embedding_from_features(const SimpleNeuralNet& net, const Features& feats)
{
    std::vector<float> emb(net.embedding_size(), 0.0f);
    check(szh_net_pooled_embedding(net.handle(), feats.data.data(), feats.len(), 0, emb.data()));
    return emb;
}

// median_embedding_from_features, lib.rs:1474-1495
inline std::vector<float> median_embedding_from_features(const SimpleNeuralNet& net, const Features& feats)
{
    std::vector<float> emb(net.embedding_size(), 0.0f);
    check(szh_net_pooled_embedding(net.handle(), feats.data.data(), feats.len(), 1, emb.data()));
    return emb;
}

// identify_speaker_from_embedding, lib.rs:1499-1529 (SIZE_MAX = no match)
inline size_t identify_speaker_from_embedding(const std::vector<float>& emb, const std::map<size_t, std::vector<float>>& speakers,
                                              float threshold)
{
    float best_sim = -3.4028235e38f;
    size_t best_id = (size_t)-1;
    for (const auto& kv : speakers) {
        const float sim = cosine_similarity(emb, kv.second);
        if (sim > best_sim) { best_sim = sim; best_id = kv.first; }
    }
    const float dyn = speakers.size() < 20 ? threshold * 0.7f : threshold;
    return best_sim > dyn ? best_id : (size_t)-1;
}

// the acceptance rule shared by identify_speaker_cosine and _feats (lib.rs:1615-1630, 1645-1660);
// SIZE_MAX = None
inline size_t best_cosine_match(const std::vector<float>& emb, const std::vector<SimpleNeuralNet::Embedding>& speakers,
                                float threshold)
{
    size_t best_idx = (size_t)-1;
    float best_val = threshold;
    for (size_t i = 0; i < speakers.size(); ++i) {
        const float mean_sim = speakers[i].second.first, std_sim = speakers[i].second.second;
        const float sim = cosine_similarity(emb, speakers[i].first);
        if (sim < mean_sim - 2.0f * std_sim) continue;
        const float adaptive = speakers.size() < 200 ? 0.3f : 1.0f;
        const float dynamic_threshold = mean_sim + std_sim * adaptive;
        const bool accepted = sim > 0.35f && (sim > dynamic_threshold || sim > 0.5f);
        if (accepted && sim > best_val) { best_val = sim; best_idx = i; }
    }
    return best_idx;
}

// identify_speaker_cosine, lib.rs:1604-1631
inline size_t identify_speaker_cosine(const SimpleNeuralNet& net, const std::vector<SimpleNeuralNet::Embedding>& speakers,
                                      const std::vector<int16_t>& sample, float threshold, const FeatureExtractor& ext)
{
    if (speakers.empty()) return (size_t)-1;
    return best_cosine_match(extract_embedding(net, sample, ext), speakers, threshold);
}

// identify_speaker_cosine_feats, lib.rs:1634-1661
inline size_t identify_speaker_cosine_feats(const SimpleNeuralNet& net, const std::vector<SimpleNeuralNet::Embedding>& speakers,
                                            const Features& windows, float threshold)
{
    if (speakers.empty()) return (size_t)-1;
    std::vector<float> emb = extract_embedding_from_features(net, windows);
    normalize(emb);
    return best_cosine_match(emb, speakers, threshold);
}

// compute_speaker_embeddings, lib.rs:1555-1599
inline std::vector<SimpleNeuralNet::Embedding> compute_speaker_embeddings(const SimpleNeuralNet& net, const FeatureExtractor& ext)
{
    std::vector<SimpleNeuralNet::Embedding> out;
    const auto lists = net.file_lists();
    const size_t d = net.embedding_size();
    for (size_t s = 0; s < net.output_size(); ++s) {
        std::vector<std::vector<float>> embeds;
        if (s < lists.size())
            for (const auto& path : lists[s]) {
                try {
                    const Features wins = load_cached_features(path, ext);
                    std::vector<float> emb = median_embedding_from_features(net, wins);
                    normalize(emb);
                    embeds.push_back(std::move(emb));
                } catch (const Error&) {}
            }
        if (embeds.empty()) { out.push_back({std::vector<float>(d, 0.0f), {0.0f, 0.0f}}); continue; }
        std::vector<float> mean(d, 0.0f);
        for (const auto& e : embeds)
            for (size_t i = 0; i < d; ++i) mean[i] += e[i];
        for (float& v : mean) v /= (float)embeds.size();
        normalize(mean);
        float msum = 0.0f;
        std::vector<float> sims;
        for (const auto& e : embeds) { sims.push_back(cosine_similarity(e, mean)); msum += sims.back(); }
        const float mean_sim = msum / (float)sims.size();
        float var = 0.0f;
        for (float x : sims) var += (x - mean_sim) * (x - mean_sim);
        var /= (float)sims.size();
        out.push_back({mean, {mean_sim, std::sqrt(var)}});
    }
    return out;
}

}  // namespace streamz
