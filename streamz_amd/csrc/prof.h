// Optional per-kernel timing with HIP events on the launch stream (off by default).
#pragma once
#include <hip/hip_runtime.h>

namespace sz {

// true while per-kernel timing is switched on
bool prof_enabled();

// Records an event pair around the launches made while the object lives, when profiling is on.
class ProfScope {
public:
    ProfScope(const char* tag, hipStream_t stream);
    ~ProfScope();
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;

private:
    int slot_ = -1;
    hipStream_t stream_ = nullptr;
};

}  // namespace sz
