// Test helper: exercises streamz_amd/csrc/npy.cpp without a GPU.
//   npy_tool write <out.npz>   writes a fixed archive the Python test reads back with numpy
//   npy_tool dump  <in.npz>    prints "name descr ndim d0 d1 .. checksum" per member
#include <cstdio>
#include <cstring>
#include <string>

#include "npy.h"

using namespace sz;

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::string err;
    if (std::strcmp(argv[1], "write") == 0) {
        NpzWriter z;
        float w[6] = {1.5f, -2.0f, 3.25f, 4.0f, 5.0f, -6.5f};
        int64_t n[1] = {44100};
        const char* txt = "a/b.wav\nc.wav";
        z.add("w1", "<f4", {2, 3}, w, sizeof(w));
        z.add("sample_rate", "<i8", {1}, n, sizeof(n));
        z.add("speaker_0_files", "|u1", {std::strlen(txt)}, txt, std::strlen(txt));
        z.add("empty", "|u1", {0}, txt, 0);
        if (!z.finish(argv[2], err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        return 0;
    }
    if (std::strcmp(argv[1], "dump") == 0) {
        std::vector<std::pair<std::string, NpyArray>> m;
        if (!npz_read(argv[2], m, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        for (auto& kv : m) {
            std::printf("%s %s %zu", kv.first.c_str(), kv.second.descr.c_str(), kv.second.shape.size());
            for (auto d : kv.second.shape) std::printf(" %llu", (unsigned long long)d);
            unsigned long long sum = 0;
            for (uint8_t b : kv.second.data) sum = sum * 131 + b;
            std::printf(" %llu\n", sum);
        }
        return 0;
    }
    return 2;
}
