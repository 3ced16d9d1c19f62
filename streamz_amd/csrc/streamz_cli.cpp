// StreamZ command-line host: the training flow of the reference binary
// (streamz-rs/src/main.rs:326-892) for 16-bit PCM WAV input at 44.1 kHz, over libstreamz_hip.so.
//
//   train_files.txt  one "path[,label]" per line (main.rs:41-64); rewritten at exit with the
//                    final labels (main.rs:870-875)
//   target_files.txt labelled lines only, written at exit (main.rs:81-89, 876)
//   model.npz        loaded when present (then the 100-epoch initial pass is skipped,
//                    main.rs:652), written at exit (main.rs:856)
//   feature_cache/   written by compute_speaker_embeddings -> load_cached_features
//
//   --eval           scores target_files.txt against the embeddings saved in model.npz
//                    (main.rs:522-625); nothing is trained or written
//   --gpus N         (no counterpart in the reference) the initial training pass runs on N GPUs, one
//                    process each: the labelled files are cut into contiguous ranges balanced on window
//                    count (szh_shard_ranges), every rank decodes and extracts only its own range, and
//                    every epoch is one optimiser step over all windows with one RCCL all-reduce of the
//                    gradients (szh_net_train_epochs_sharded) instead of the per-file batches of 8 of
//                    train_from_feature_map: 100 optimiser steps in all where the reference takes
//                    100 x windows / 8 per file, so the model of this pass is far less trained than the
//                    reference's -- it is the throughput schedule, not a replacement for the default one.
//                    The labelling pass then runs on the first GPU (which extracts the remaining files).
//   --share-gpu      with --gpus: the ranks take the GPUs round-robin and sum on the host (a rehearsal of
//                    the multi-rank flow on a machine with fewer GPUs than ranks)
//
// Not handled here (outside the accelerated path): MP3 input and cache/*.wav,
// --encode / --decode / --checksum.  The reference processes the second pass under a
// rayon par_iter whose iterations serialise on the net's write lock in an unspecified order;
// here the files are taken in list order, one valid schedule of that loop.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include <poll.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "streamz.hpp"

using namespace streamz;

namespace {

const char* MODEL_PATH = "model.npz";              // main.rs:21
const char* TRAIN_FILE_LIST = "train_files.txt";   // main.rs:22
const char* TARGET_FILE_LIST = "target_files.txt"; // main.rs:23
const float DEFAULT_CONF_THRESHOLD = 0.8f;         // main.rs:27
const float DEFAULT_BURN_IN_FRAC = 0.2f;           // main.rs:30
const size_t TRAIN_EPOCHS = 100;                   // main.rs:33
const float DROPOUT_PROB = DEFAULT_DROPOUT;        // main.rs:35
const size_t BATCH_SIZE = 8;                       // main.rs:37

struct Entry { std::string path; bool labelled; size_t cls; };

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

bool parse_usize(const std::string& s, size_t& out)
{
    if (s.empty()) return false;
    size_t i = 0;
    if (s[0] == '+') i = 1;  // Rust's usize::from_str accepts a leading '+'
    if (i >= s.size()) return false;
    size_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (size_t)(s[i] - '0');
    }
    out = v;
    return true;
}

// load_train_files, main.rs:41-64
std::vector<Entry> load_train_files(const char* path)
{
    std::vector<Entry> files;
    std::ifstream in(path);
    if (!in) return files;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<std::string> parts;
        size_t pos = 0;
        while (true) {
            const size_t c = line.find(',', pos);
            parts.push_back(line.substr(pos, c == std::string::npos ? std::string::npos : c - pos));
            if (c == std::string::npos) break;
            pos = c + 1;
        }
        const std::string p = trim(parts[0]);
        if (p.empty()) continue;
        Entry e{p, false, 0};
        if (parts.size() > 1) {
            size_t cls;
            if (parse_usize(trim(parts[1]), cls)) { e.labelled = true; e.cls = cls; }
        }
        files.push_back(e);
    }
    return files;
}

// write_train_files, main.rs:66-79 / write_target_files, main.rs:81-89
void write_list(const char* path, const std::vector<Entry>& files, bool labelled_only)
{
    std::ofstream out(path);
    if (!out) return;
    for (const Entry& e : files) {
        if (e.labelled) out << e.path << "," << e.cls << "\n";
        else if (!labelled_only) out << e.path << "\n";
    }
}

// count_speakers, main.rs:129-135
size_t count_speakers(const std::vector<Entry>& files)
{
    std::set<size_t> s;
    for (const Entry& e : files)
        if (e.labelled) s.insert(e.cls);
    return s.size();
}

bool exists(const char* p)
{
    struct stat st;
    return ::stat(p, &st) == 0;
}

int find_arg(int argc, char** argv, const char* name)
{
    for (int i = 1; i < argc; ++i)
        if (std::strcmp(argv[i], name) == 0) return i;
    return -1;
}

// recompute_embeddings, main.rs:216-241: every speaker's centroid again from the embeddings collected
// for it so far.  (The second pass already refreshes a speaker's centroid after each of its files,
// so this leaves the map as it is; it is kept because the reference calls it.)
void recompute_embeddings(const std::map<size_t, std::vector<std::vector<float>>>& speaker_features,
                          std::map<size_t, std::vector<float>>& speaker_embeddings)
{
    for (const auto& kv : speaker_features) speaker_embeddings[kv.first] = average_vectors(kv.second);
}

// print_embedding_quality, main.rs:243-279
void print_embedding_quality(const SimpleNeuralNet& net, const FeatureExtractor& ext)
{
    std::vector<SimpleNeuralNet::Embedding> e = net.embeddings();
    if (!e.empty()) std::printf("Saved embeddings found in model.npz:\n");
    else e = compute_speaker_embeddings(net, ext);
    if (e.empty()) { std::printf("No embeddings available to evaluate\n"); return; }
    float sum = 0.0f;
    for (size_t i = 0; i < e.size(); ++i) {
        sum += e[i].second.first;
        std::printf("Speaker %zu: mean similarity %.4f, std dev %.4f\n", i, e[i].second.first, e[i].second.second);
    }
    std::printf("Average mean similarity: %.4f\n", sum / (float)e.size());
}

// --eval, main.rs:522-625: labels of both lists are renumbered densely in ascending order
// (build_label_map / normalize_with_map, main.rs:281-304); every target file's mean embedding
// is compared with the saved speaker embeddings; the best similarity above the threshold wins.
int run_eval(const std::vector<Entry>& train_files, const std::vector<Entry>& target_files,
             const std::map<std::string, Features>& feature_map, float conf_threshold)
{
    std::printf("Evaluating with threshold = %g\n", conf_threshold);
    std::set<size_t> labels;
    for (const Entry& e : train_files) if (e.labelled) labels.insert(e.cls);
    for (const Entry& e : target_files) labels.insert(e.cls);
    std::map<size_t, size_t> label_map;
    for (size_t v : labels) { const size_t id = label_map.size(); label_map[v] = id; }
    if (!exists(MODEL_PATH)) {
        std::fprintf(stderr, "Model file %s not found. Please train first.\n", MODEL_PATH);
        return 0;
    }
    std::printf("Loading model from %s\n", MODEL_PATH);
    SimpleNeuralNet net = [&]() -> SimpleNeuralNet {
        try { return SimpleNeuralNet::load(MODEL_PATH); }
        catch (const Error& e) { std::fprintf(stderr, "Failed to load model: %s\n", e.what()); std::exit(0); }
    }();
    const std::vector<SimpleNeuralNet::Embedding> saved = net.embeddings();
    std::printf("Model contains %zu saved embeddings\n", saved.size());
    std::fprintf(stderr, "Total speaker embeddings available: %zu\n", saved.size());
    size_t true_positive = 0, false_positive = 0, false_negative = 0, correct = 0;
    const size_t none = (size_t)-1;
    for (const Entry& e : target_files) {
        const size_t true_class = label_map[e.cls];
        auto it = feature_map.find(e.path);
        if (it == feature_map.end()) { std::fprintf(stderr, "No features found for %s\n", e.path.c_str()); continue; }
        std::vector<float> embedding = extract_embedding_from_features(net, it->second);
        normalize(embedding);
        float norm = 0.0f;
        for (float v : embedding) norm += v * v;
        std::fprintf(stderr, "\nEvaluating file: %s\nTrue class: %zu\nEmbedding norm: %.6f\n", e.path.c_str(), true_class,
                     std::sqrt(norm));
        size_t best_id = none;
        float best_sim = -3.4028235e38f;
        for (size_t id = 0; id < saved.size(); ++id) {
            const float sim = cosine_similarity(embedding, saved[id].first);
            std::fprintf(stderr, "  -> Similarity to speaker %zu: %.6f\n", id, sim);
            if (sim > conf_threshold && sim > best_sim) { best_sim = sim; best_id = id; }
        }
        if (best_id == true_class) { ++correct; ++true_positive; }
        else if (best_id == none) { ++false_negative; std::fprintf(stderr, "  -> Unclassified\n"); }
        else {
            ++false_positive;
            std::fprintf(stderr, "  -> Misclassified: predicted speaker %zu, true speaker %zu\n", best_id, true_class);
        }
    }
    const float total = (float)std::max<size_t>(target_files.size(), 1);
    const float accuracy = (float)correct / total;
    const float precision = (float)true_positive / (float)std::max<size_t>(true_positive + false_positive, 1);
    const float recall = (float)true_positive / (float)std::max<size_t>(true_positive + false_negative, 1);
    const float f1 = 2.0f * precision * recall / std::max(precision + recall, 1e-6f);
    std::printf("\nEvaluation complete:\n");
    std::printf("  Accuracy:  %.2f%%\n", 100.0f * accuracy);
    std::printf("  Precision: %.2f%%\n", 100.0f * precision);
    std::printf("  Recall:    %.2f%%\n", 100.0f * recall);
    std::printf("  F1-score:  %.2f%%\n", 100.0f * f1);
    return 0;
}

// The ranks of --gpus N: this process is rank 0, the others are forked children.  Two pipes per child: the
// child reports over one that it is ready (device selected, its files extracted) or that it failed, and
// receives the communicator id over the other.  Every way out of main reaps the children.
struct Ranks {
    int rank = 0, world = 1;
    int id_in = -1, status_out = -1;       // a child's ends
    std::vector<int> id_out, status_in;     // rank 0's ends, one per child
    std::vector<pid_t> children;
    std::vector<bool> reaped;

    void spawn(int n)
    {
        world = n;
        ::signal(SIGPIPE, SIG_IGN);  // a rank that died shows as a failed write, not as a signal
        for (int r = 1; r < world && rank == 0; ++r) {
            int to_child[2], from_child[2];
            if (::pipe(to_child) != 0 || ::pipe(from_child) != 0) throw Error(SZH_EINVAL, "pipe failed");
            std::fflush(nullptr);
            const pid_t pid = ::fork();
            if (pid < 0) throw Error(SZH_EINVAL, "fork failed");
            if (pid == 0) {
                rank = r; id_in = to_child[0]; status_out = from_child[1];
                ::close(to_child[1]); ::close(from_child[0]);
                for (int fd : id_out) ::close(fd);
                for (int fd : status_in) ::close(fd);
                id_out.clear(); status_in.clear(); children.clear(); reaped.clear();
                if (!std::freopen("/dev/null", "w", stdout)) {}  // messages come from rank 0 only
            } else {
                ::close(to_child[0]); ::close(from_child[1]);
                id_out.push_back(to_child[1]); status_in.push_back(from_child[0]);
                children.push_back(pid); reaped.push_back(false);
            }
        }
    }
    // a child: tell rank 0 how the set-up went
    void report(char status) { if (status_out >= 0 && ::write(status_out, &status, 1) != 1) {} }
    // rank 0: every child must report 'R'; a child that exits or stays silent for ten minutes is a failure
    void await_ready()
    {
        for (size_t c = 0; c < children.size(); ++c) {
            char st = 0;
            for (int waited_ms = 0;; waited_ms += 200) {
                struct pollfd pf = {status_in[c], POLLIN, 0};
                const int pr = ::poll(&pf, 1, 200);
                if (pr > 0) { if (::read(status_in[c], &st, 1) != 1) st = 0; break; }
                int status = 0;
                if (::waitpid(children[c], &status, WNOHANG) == children[c]) { reaped[c] = true; break; }
                if (waited_ms > 600000) break;
            }
            if (st != 'R') throw Error(SZH_EINVAL, "rank " + std::to_string(c + 1) + " of --gpus did not come up (see its messages above)");
        }
    }
    // rank 0 after the pass: all children must have ended well
    void join()
    {
        for (int fd : id_out) ::close(fd);
        id_out.clear();
        bool ok = true;
        for (size_t c = 0; c < children.size(); ++c) {
            if (reaped[c]) { ok = false; continue; }
            int status = 0;
            if (::waitpid(children[c], &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) ok = false;
            reaped[c] = true;
        }
        if (!ok) throw Error(SZH_EINVAL, "a rank of the initial pass failed");
    }
    ~Ranks()
    {
        for (int fd : id_out) ::close(fd);  // a child still waiting for the id sees the end of the pipe and leaves
        for (int fd : status_in) ::close(fd);
        for (size_t c = 0; c < children.size(); ++c) {
            if (reaped[c]) continue;
            int status = 0;
            if (::waitpid(children[c], &status, WNOHANG) == 0) {  // still running on an exit path that did not join: stop it
                ::kill(children[c], SIGTERM);
                ::waitpid(children[c], &status, 0);
            }
        }
    }
};

// decode + extract the listed paths into feature_map (batch_resample, lib.rs:541-547: unreadable files are
// dropped; then the file-parallel extraction loop, main.rs:500-508)
void extract_into(const FeatureExtractor& extractor, const std::vector<std::string>& paths, std::map<std::string, Features>& feature_map)
{
    auto audio = batch_resample(paths);
    std::vector<std::vector<int16_t>> clips;
    for (auto& a : audio) clips.push_back(std::move(a.second));
    std::vector<Features> feats = extractor.extract_batch(clips);
    for (size_t i = 0; i < audio.size(); ++i) feature_map[audio[i].first] = std::move(feats[i]);
}

// this rank's contiguous range of the labelled files, balanced on the window counts their headers give
std::pair<size_t, size_t> rank_range(const std::vector<std::pair<std::string, size_t>>& refs, int rank, int world)
{
    std::vector<uint64_t> windows(refs.size(), 0);
    for (size_t f = 0; f < refs.size(); ++f) {
        uint64_t frames = 0;
        uint32_t rate = 0;
        if (szh_wav_info(refs[f].first.c_str(), &frames, &rate, nullptr) == SZH_OK && rate == DEFAULT_SAMPLE_RATE)
            windows[f] = szh_window_count(frames);  // a file that cannot be read brings no windows (and is skipped later)
    }
    std::vector<uint32_t> cut((size_t)world + 1);
    check(szh_shard_ranges(windows.data(), (uint32_t)refs.size(), (uint32_t)world, cut.data()));
    return {cut[(size_t)rank], cut[(size_t)rank + 1]};
}

// The initial pass of one rank over files [lo, hi) of refs.
float sharded_initial_pass(SimpleNeuralNet& net, const std::vector<std::pair<std::string, size_t>>& refs, size_t lo, size_t hi,
                           const std::map<std::string, Features>& feature_map, Ranks& ranks, uint64_t seed)
{
    uint8_t id[SZH_COMM_ID_BYTES];
    if (ranks.rank == 0) {
        ranks.await_ready();
        check(szh_comm_unique_id(id));
        for (int fd : ranks.id_out)
            if (::write(fd, id, sizeof(id)) != (ssize_t)sizeof(id)) throw Error(SZH_EINVAL, "could not hand the communicator id to a rank");
    } else {
        ranks.report('R');
        if (::read(ranks.id_in, id, sizeof(id)) != (ssize_t)sizeof(id)) throw Error(SZH_EINVAL, "no communicator id from rank 0");
    }
    szh_comm* comm = nullptr;
    check(szh_comm_create(id, ranks.rank, ranks.world, &comm));
    struct CommGuard { szh_comm* c; ~CommGuard() { szh_comm_destroy(c); } } guard{comm};
    check(szh_net_broadcast_params_dev(net.handle(), comm, 0, nullptr));
    std::vector<float> x;
    std::vector<int32_t> labels;
    for (size_t f = lo; f < hi; ++f) {
        auto it = feature_map.find(refs[f].first);
        if (it == feature_map.end()) continue;
        x.insert(x.end(), it->second.data.begin(), it->second.data.end());
        labels.insert(labels.end(), it->second.len(), (int32_t)refs[f].second);
    }
    float loss = 0.0f;
    check(szh_net_train_epochs_sharded(net.handle(), comm, x.data(), labels.data(), labels.size(), (uint32_t)TRAIN_EPOCHS, 0.01f,
                                       DROPOUT_PROB, seed, &loss));
    return loss;
}

}  // namespace

int main(int argc, char** argv)
{
    Ranks ranks;  // outlives everything below: its destructor reaps the forked ranks on every way out
    try {
        float conf_threshold = DEFAULT_CONF_THRESHOLD;
        bool have_burn = false;
        size_t burn_in_limit = 0;
        int idx;
        uint64_t seed = 0x5EEDull;
        if ((idx = find_arg(argc, argv, "--seed")) >= 0 && idx + 1 < argc) seed = std::strtoull(argv[idx + 1], nullptr, 10);
        const bool eval_mode = find_arg(argc, argv, "--eval") >= 0;  // main.rs:350
        for (const char* flag : {"--encode", "--decode", "--checksum"})
            if (find_arg(argc, argv, flag) >= 0) {
                std::fprintf(stderr, "%s is not part of the accelerated path and is not handled by this binary\n", flag);
                return 2;
            }
        int device = -1;
        if ((idx = find_arg(argc, argv, "--device")) >= 0 && idx + 1 < argc) device = std::atoi(argv[idx + 1]);
        if (find_arg(argc, argv, "--check-embeddings") >= 0) {  // main.rs:357-366
            if (device >= 0) check(szh_init(device));
            FeatureExtractor extractor;
            try {
                SimpleNeuralNet net = SimpleNeuralNet::load(MODEL_PATH);
                std::printf("Loaded %s for embedding check\n", MODEL_PATH);
                print_embedding_quality(net, extractor);
            } catch (const Error& e) {
                std::fprintf(stderr, "Failed to load model from %s: %s\n", MODEL_PATH, e.what());
            }
            return 0;
        }
        if ((idx = find_arg(argc, argv, "--threshold")) >= 0) {  // main.rs:367-382
            if (idx + 1 < argc) {
                char* end = nullptr;
                const float v = std::strtof(argv[idx + 1], &end);
                if (end && *end == 0 && end != argv[idx + 1]) conf_threshold = v;
                else std::fprintf(stderr, "Invalid value for --threshold '%s', using default %g\n", argv[idx + 1], DEFAULT_CONF_THRESHOLD);
            } else {
                std::fprintf(stderr, "Missing value for --threshold, using default %g\n", DEFAULT_CONF_THRESHOLD);
            }
        }
        if ((idx = find_arg(argc, argv, "--burn-in-limit")) >= 0) {  // main.rs:398-410
            size_t v;
            if (idx + 1 < argc && parse_usize(argv[idx + 1], v)) { have_burn = true; burn_in_limit = v; }
            else std::fprintf(stderr, "Invalid or missing value for --burn-in-limit, using automatic setting\n");
        }

        std::vector<Entry> train_files = load_train_files(TRAIN_FILE_LIST);  // main.rs:471-475
        if (train_files.empty()) {
            std::fprintf(stderr, "%s is empty\n", TRAIN_FILE_LIST);
            return 0;
        }
        const std::vector<Entry> original = train_files;
        const bool model_exists = exists(MODEL_PATH);

        // --gpus N: the ranks are separate processes, forked here: after every early way out above, before
        // anything touches a GPU
        int world = 1;
        bool sharded = false;  // --gpus given (1 is allowed: the same schedule on one rank)
        const bool share_gpu = find_arg(argc, argv, "--share-gpu") >= 0;
        if ((idx = find_arg(argc, argv, "--gpus")) >= 0 && idx + 1 < argc) { world = std::max(1, std::atoi(argv[idx + 1])); sharded = true; }
        if (sharded && (eval_mode || model_exists)) {
            std::fprintf(stderr, "--gpus only spreads the initial training pass, which this run does not make; using one GPU\n");
            world = 1;
            sharded = false;
        }
        if (sharded && share_gpu && world > 1) ::setenv("STREAMZ_COMM", "host", 1);  // RCCL refuses ranks that share a device
        if (sharded) ranks.spawn(world);
        const int rank = ranks.rank;
        try {
            if (sharded) {
                // (devices are counted here, after the fork: no GPU state crosses it)
                const int n_dev = share_gpu ? std::max(1, szh_device_count()) : 1;
                check(szh_init(share_gpu ? rank % n_dev : rank));
            } else if (device >= 0) {
                check(szh_init(device));
            }
        } catch (const Error&) {
            ranks.report('F');
            throw;
        }
        FeatureExtractor extractor;

        size_t num_speakers = count_speakers(train_files);  // main.rs:627-650
        if (!eval_mode && !model_exists && num_speakers == 0) {
            num_speakers = 1;
            train_files[0].labelled = true;
            train_files[0].cls = 0;
            std::printf("No labeled speakers found \xe2\x80\x94 assigned speaker 0 to first file.\n");
        }
        std::vector<std::pair<std::string, size_t>> refs;  // the labelled files, in list order (main.rs:653-657)
        for (const Entry& e : train_files)
            if (e.labelled) refs.emplace_back(e.path, e.cls);

        // decode once (main.rs:486-490), then the file-parallel extraction loop (main.rs:500-508); with --gpus
        // a rank takes only its own range of the labelled files, and rank 0 the rest after the initial pass
        std::vector<std::string> path_list;
        for (const Entry& e : train_files) path_list.push_back(e.path);
        std::vector<Entry> target_files;
        if (eval_mode) {  // main.rs:477, 487-489; load_target_files keeps labelled lines only (main.rs:91-111)
            for (const Entry& e : load_train_files(TARGET_FILE_LIST))
                if (e.labelled) { target_files.push_back(e); path_list.push_back(e.path); }
        }
        std::map<std::string, Features> feature_map;
        std::pair<size_t, size_t> range{0, refs.size()};
        try {
            if (sharded) {
                range = rank_range(refs, rank, world);
                std::vector<std::string> mine;
                for (size_t f = range.first; f < range.second; ++f) mine.push_back(refs[f].first);
                extract_into(extractor, mine, feature_map);
            } else {
                extract_into(extractor, path_list, feature_map);
            }
        } catch (const Error&) {
            ranks.report('F');
            throw;
        }

        const size_t dataset_size = train_files.size();  // main.rs:517-519
        size_t burn_default = (size_t)std::ceil((float)dataset_size * DEFAULT_BURN_IN_FRAC);
        burn_default = std::min<size_t>(std::max<size_t>(burn_default, 10), 50);
        const size_t burn_in_limit_val = have_burn ? burn_in_limit : burn_default;

        if (eval_mode) {
            for (const Entry& e : train_files)
                if (!feature_map.count(e.path)) std::fprintf(stderr, "No features found for training path: %s\n", e.path.c_str());
            return run_eval(train_files, target_files, feature_map, conf_threshold);
        }

        SimpleNeuralNet net = [&]() -> SimpleNeuralNet {
            if (model_exists) {
                try {
                    SimpleNeuralNet n = SimpleNeuralNet::load(MODEL_PATH);
                    std::printf("Loaded saved model from %s\n", MODEL_PATH);
                    n.set_embeddings(compute_speaker_embeddings(n, extractor));
                    return n;
                } catch (const Error& e) {
                    std::fprintf(stderr, "Failed to load model: %s\n", e.what());
                }
            }
            return SimpleNeuralNet(FEATURE_SIZE, 512, 256, std::max<size_t>(num_speakers, 1), seed);
        }();

        if (!model_exists) {  // main.rs:652-669
            if (sharded) {
                const float loss = sharded_initial_pass(net, refs, range.first, range.second, feature_map, ranks, seed);
                if (rank != 0) {  // a rank's part ends with the initial pass
                    check(szh_shutdown());
                    return 0;
                }
                ranks.join();
                std::printf("Initial training loss: %.4f  (%d GPU%s, one optimiser step per epoch)\n", loss, world, world > 1 ? "s" : "");
                // the labelling pass runs here alone and walks every file: extract what the other ranks held
                std::vector<std::string> rest;
                for (const std::string& p : path_list)
                    if (!feature_map.count(p)) rest.push_back(p);
                extract_into(extractor, rest, feature_map);
                for (const auto& fc : refs)  // as train_from_feature_map does for every file it trained on (lib.rs:655)
                    if (feature_map.count(fc.first)) net.record_training_file(fc.second, fc.first);
            } else if (!refs.empty()) {
                const float loss = train_from_feature_map(net, feature_map, refs, TRAIN_EPOCHS, 0.01f, DROPOUT_PROB, BATCH_SIZE, seed);
                std::printf("Initial training loss: %.4f\n", loss);
            }
        }
        for (const Entry& e : train_files)
            if (!feature_map.count(e.path)) std::fprintf(stderr, "No features found for training path: %s\n", e.path.c_str());

        // second pass, main.rs:710-835
        std::map<size_t, std::vector<std::vector<float>>> speaker_features;
        std::map<size_t, std::vector<float>> speaker_embeddings;
        {
            const auto embeds = compute_speaker_embeddings(net, extractor);
            for (size_t i = 0; i < embeds.size(); ++i) speaker_embeddings[i] = embeds[i].first;
        }
        float total_loss = 0.0f;
        size_t loss_count = 0;
        for (Entry& e : train_files) {
            auto it = feature_map.find(e.path);
            if (it == feature_map.end()) { std::fprintf(stderr, "Missing audio for %s\n", e.path.c_str()); continue; }
            const Features& windows = it->second;
            if (windows.len() < 5) { std::fprintf(stderr, "Skipping %s, too short\n", e.path.c_str()); continue; }
            std::vector<float> emb = extract_embedding_from_features(net, windows);
            normalize(emb);
            const size_t count = loss_count;
            const bool burn_phase = count < burn_in_limit_val;
            const float threshold = burn_phase ? 0.5f : DEFAULT_CONF_THRESHOLD;
            size_t speaker_id;
            if (burn_phase && !e.labelled) {
                speaker_id = net.output_size();
                net.add_output_class(seed + 7919 * (loss_count + 1));
                net.record_training_file(speaker_id, e.path);
                e.labelled = true;
                e.cls = speaker_id;
            } else if (e.labelled) {
                speaker_id = e.cls;
            } else {
                size_t matched = identify_speaker_from_embedding(emb, speaker_embeddings, threshold);
                if (matched >= net.output_size()) {
                    net.add_output_class(seed + 7919 * (loss_count + 1));
                    matched = net.output_size() - 1;
                }
                e.labelled = true;
                e.cls = matched;
                speaker_id = matched;
            }
            const float lr = count < 1000 ? 0.05f : 0.01f;
            const float loss = pretrain_from_features(net, windows, speaker_id, net.output_size(), 5, lr, DROPOUT_PROB,
                                                      BATCH_SIZE, seed + 104729 * (loss_count + 1));
            net.record_training_file(speaker_id, e.path);
            total_loss += loss;
            speaker_features[speaker_id].push_back(emb);
            speaker_embeddings[speaker_id] = average_vectors(speaker_features[speaker_id]);
            loss_count += 1;
            if (loss_count % 100 == 0) recompute_embeddings(speaker_features, speaker_embeddings);  // main.rs:826-829
        }

        // main.rs:845-891
        const auto new_embeddings = compute_speaker_embeddings(net, extractor);
        for (size_t i = 0; i < new_embeddings.size(); ++i) {
            float norm = 0.0f;
            for (float x : new_embeddings[i].first) norm += x * x;
            std::printf("Saving Speaker %zu \xe2\x86\x92 mean_sim: %.4f, std_sim: %.4f, norm: %.4f\n", i,
                        new_embeddings[i].second.first, new_embeddings[i].second.second, std::sqrt(norm));
        }
        net.set_embeddings(new_embeddings);
        try { net.save(MODEL_PATH); } catch (const Error& e) { std::fprintf(stderr, "Failed to save model: %s\n", e.what()); }
        std::printf("Computed %zu embeddings for %zu speakers\n", net.embeddings().size(), net.output_size());
        if (loss_count > 0) std::printf("Average training loss: %.4f\n", total_loss / (float)loss_count);

        std::vector<Entry> updated = original;  // original paths + final labels
        for (size_t i = 0; i < updated.size(); ++i) { updated[i].labelled = train_files[i].labelled; updated[i].cls = train_files[i].cls; }
        write_list(TRAIN_FILE_LIST, updated, false);
        write_list(TARGET_FILE_LIST, train_files, true);
        std::printf("Updated training file labels:\n");
        for (const Entry& e : updated) {
            if (e.labelled) std::printf("%s -> speaker %zu\n", e.path.c_str(), e.cls + 1);
            else std::printf("%s -> speaker unknown\n", e.path.c_str());
        }
        std::printf("Processed %zu speakers in this batch.\n", count_speakers(train_files));
        std::printf("Number of speakers discovered: %zu\n", net.output_size());
        for (size_t i = 0; i < net.output_size(); ++i)
            std::printf("Speaker %zu: %zu samples\n", i, speaker_features.count(i) ? speaker_features[i].size() : (size_t)0);
        return 0;
    } catch (const Error& e) {
        std::fprintf(stderr, "StreamZ: %s\n", e.what());
        return 1;
    }
}
