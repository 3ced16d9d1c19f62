// Minimal .npy / .npz reader and writer for the two on-disk formats of the reference:
// feature_cache/*.npy (lib.rs:564, 576) and model.npz (lib.rs:1081-1281).
#pragma once
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

namespace sz {

struct NpyArray {
    std::string descr;            // "<f4", "<i8", "|u1", ...
    std::vector<uint64_t> shape;
    bool fortran_order = false;
    std::vector<uint8_t> data;    // raw little-endian payload
    uint64_t count() const { uint64_t c = 1; for (uint64_t s : shape) c *= s; return c; }
};

// NPY v1.0 image of an array (header padded so that the payload starts on a 64-byte boundary)
void npy_serialize(const std::string& descr, const std::vector<uint64_t>& shape, const void* data, size_t bytes,
                   std::vector<uint8_t>& out);
// false + err on a malformed image
bool npy_parse(const uint8_t* buf, size_t len, NpyArray& out, std::string& err);
bool npy_read_file(const std::string& path, NpyArray& out, std::string& err);
bool npy_write_file(const std::string& path, const std::string& descr, const std::vector<uint64_t>& shape,
                    const void* data, size_t bytes, std::string& err);

// .npz = zip archive of "<name>.npy" members, written uncompressed (stored) like
// ndarray-npy's NpzWriter::new; the reader also inflates deflated members (numpy's
// savez_compressed) and strips the ".npy" suffix from member names.
class NpzWriter {
public:
    void add(const std::string& name, const std::string& descr, const std::vector<uint64_t>& shape, const void* data,
             size_t bytes);
    bool finish(const std::string& path, std::string& err);

private:
    struct Member { std::string file; uint32_t crc; uint32_t size; uint32_t offset; };
    std::vector<uint8_t> blob_;
    std::vector<Member> members_;
};

bool npz_read(const std::string& path, std::vector<std::pair<std::string, NpyArray>>& members, std::string& err);

}  // namespace sz
