#include "prof.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "internal.h"
#include "streamz_hip.h"

namespace sz {

namespace {
struct Pair { const char* tag; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Pair> g_pairs;          // pairs recorded since the last report
std::vector<Pair> g_free;           // recycled events
}  // namespace

bool prof_enabled() { return g_on; }

ProfScope::ProfScope(const char* tag, hipStream_t stream)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lock(g_mu);
    Pair p{tag, nullptr, nullptr};
    if (!g_free.empty()) { p = g_free.back(); g_free.pop_back(); p.tag = tag; }
    else if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
    if (hipEventRecord(p.a, stream) != hipSuccess) return;
    g_pairs.push_back(p);
    slot_ = (int)g_pairs.size() - 1;
    stream_ = stream;
}

ProfScope::~ProfScope()
{
    if (slot_ < 0) return;
    std::lock_guard<std::mutex> lock(g_mu);
    if ((size_t)slot_ < g_pairs.size()) (void)hipEventRecord(g_pairs[slot_].b, stream_);
}

}  // namespace sz

using namespace sz;

extern "C" {

int szh_profile_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_on = on != 0;
    return SZH_OK;
}

int64_t szh_profile_report(char* buf, uint64_t capacity)
{
    if (hipDeviceSynchronize() != hipSuccess) { set_error("device synchronize failed"); return SZH_ENODEVICE; }
    std::lock_guard<std::mutex> lock(g_mu);
    std::map<std::string, std::pair<uint64_t, double>> agg;
    for (Pair& p : g_pairs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto& e = agg[p.tag];
            e.first += 1;
            e.second += ms;
        }
        g_free.push_back(p);
    }
    g_pairs.clear();
    std::string text;
    char line[160];
    for (auto& kv : agg) {
        std::snprintf(line, sizeof(line), "%s %llu %.6f\n", kv.first.c_str(), (unsigned long long)kv.second.first,
                      kv.second.second);
        text += line;
    }
    if (buf && capacity) {
        const size_t c = text.size() < capacity - 1 ? text.size() : (size_t)capacity - 1;
        std::memcpy(buf, text.data(), c);
        buf[c] = 0;
    }
    return (int64_t)text.size();
}

}  // extern "C"
