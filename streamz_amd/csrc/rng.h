// Counter-based random numbers shared by the host-side shuffles and the device-side dropout.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SZ_RNG_FN __host__ __device__ inline
#else
#define SZ_RNG_FN inline
#endif

namespace sz {

SZ_RNG_FN uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// 24-bit uniform in [0, 1), like rand's gen::<f32>()
SZ_RNG_FN float uniform01(uint64_t seed, uint64_t a, uint64_t b)
{
    const uint64_t h = mix64(mix64(seed ^ (a * 0xD1342543DE82EF95ull)) + b);
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}

}  // namespace sz
