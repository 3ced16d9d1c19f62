#include "npy.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sz {

namespace {

void put16(std::vector<uint8_t>& v, uint16_t x) { v.push_back(x & 0xff); v.push_back(x >> 8); }
void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((x >> (8 * i)) & 0xff); }
uint16_t get16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

bool read_all(const std::string& path, std::vector<uint8_t>& out, std::string& err)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); err = "cannot size " + path; return false; }
    out.resize((size_t)n);
    const size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    if (got != (size_t)n) { err = "short read on " + path; return false; }
    return true;
}

bool write_all(const std::string& path, const std::vector<uint8_t>& data, std::string& err)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    const size_t put = data.empty() ? 0 : std::fwrite(data.data(), 1, data.size(), f);
    const bool ok = put == data.size() && std::fclose(f) == 0;
    if (!ok) err = "short write on " + path;
    return ok;
}

}  // namespace

void npy_serialize(const std::string& descr, const std::vector<uint64_t>& shape, const void* data, size_t bytes,
                   std::vector<uint8_t>& out)
{
    std::string dict = "{'descr': '" + descr + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); ++i) {
        dict += std::to_string(shape[i]);
        if (shape.size() == 1 || i + 1 < shape.size()) dict += ",";
        if (i + 1 < shape.size()) dict += " ";
    }
    dict += "), }";
    size_t total = 10 + dict.size() + 1;
    const size_t pad = (64 - total % 64) % 64;
    dict.append(pad, ' ');
    dict += "\n";
    out.clear();
    const uint8_t magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    out.insert(out.end(), magic, magic + 8);
    put16(out, (uint16_t)dict.size());
    out.insert(out.end(), dict.begin(), dict.end());
    const uint8_t* p = static_cast<const uint8_t*>(data);
    out.insert(out.end(), p, p + bytes);
}

bool npy_parse(const uint8_t* buf, size_t len, NpyArray& out, std::string& err)
{
    if (len < 10 || buf[0] != 0x93 || std::memcmp(buf + 1, "NUMPY", 5) != 0) { err = "not an NPY file"; return false; }
    const int major = buf[6];
    size_t hlen, hoff;
    if (major == 1) { hlen = get16(buf + 8); hoff = 10; }
    else if (major == 2 || major == 3) { if (len < 12) { err = "truncated NPY header"; return false; } hlen = get32(buf + 8); hoff = 12; }
    else { err = "unsupported NPY version"; return false; }
    if (hoff + hlen > len) { err = "truncated NPY header"; return false; }
    const std::string h(reinterpret_cast<const char*>(buf + hoff), hlen);
    auto find_val = [&](const char* key) -> size_t {
        size_t p = h.find(key);
        if (p == std::string::npos) return p;
        p = h.find(':', p);
        return p == std::string::npos ? p : p + 1;
    };
    size_t p = find_val("'descr'");
    if (p == std::string::npos) { err = "NPY header lacks descr"; return false; }
    size_t q1 = h.find('\'', p);
    size_t q2 = q1 == std::string::npos ? q1 : h.find('\'', q1 + 1);
    if (q2 == std::string::npos) { err = "NPY descr is not a simple dtype"; return false; }
    out.descr = h.substr(q1 + 1, q2 - q1 - 1);
    p = find_val("'fortran_order'");
    if (p == std::string::npos) { err = "NPY header lacks fortran_order"; return false; }
    out.fortran_order = h.find("True", p) != std::string::npos && h.find("True", p) < h.find(',', p);
    p = find_val("'shape'");
    if (p == std::string::npos) { err = "NPY header lacks shape"; return false; }
    size_t b1 = h.find('(', p), b2 = h.find(')', p);
    if (b1 == std::string::npos || b2 == std::string::npos || b2 < b1) { err = "bad NPY shape"; return false; }
    out.shape.clear();
    uint64_t cur = 0;
    bool have = false;
    for (size_t i = b1 + 1; i < b2; ++i) {
        const char c = h[i];
        if (c >= '0' && c <= '9') { cur = cur * 10 + (uint64_t)(c - '0'); have = true; }
        else if (have) { out.shape.push_back(cur); cur = 0; have = false; }
    }
    if (have) out.shape.push_back(cur);
    size_t item = 0;
    if (out.descr.size() >= 3) item = (size_t)std::atoi(out.descr.c_str() + 2);
    if (item == 0) { err = "unsupported NPY dtype " + out.descr; return false; }
    // the element count and the byte size are multiplied out with an overflow check: a header can claim any shape
    uint64_t need = item;
    for (uint64_t d : out.shape) {
        if (d != 0 && need > UINT64_MAX / d) { err = "NPY shape overflows"; return false; }
        need *= d;
    }
    if (need > len || hoff + hlen > len - need) { err = "truncated NPY payload"; return false; }
    out.data.assign(buf + hoff + hlen, buf + hoff + hlen + need);
    return true;
}

bool npy_read_file(const std::string& path, NpyArray& out, std::string& err)
{
    std::vector<uint8_t> buf;
    if (!read_all(path, buf, err)) return false;
    return npy_parse(buf.data(), buf.size(), out, err);
}

bool npy_write_file(const std::string& path, const std::string& descr, const std::vector<uint64_t>& shape,
                    const void* data, size_t bytes, std::string& err)
{
    std::vector<uint8_t> img;
    npy_serialize(descr, shape, data, bytes, img);
    return write_all(path, img, err);
}

void NpzWriter::add(const std::string& name, const std::string& descr, const std::vector<uint64_t>& shape,
                    const void* data, size_t bytes)
{
    std::vector<uint8_t> img;
    npy_serialize(descr, shape, data, bytes, img);
    Member m;
    m.file = name + ".npy";
    m.crc = (uint32_t)crc32(0L, img.data(), (uInt)img.size());
    m.size = (uint32_t)img.size();
    m.offset = (uint32_t)blob_.size();
    put32(blob_, 0x04034b50u);  // local file header
    put16(blob_, 20);           // version needed
    put16(blob_, 0);            // flags
    put16(blob_, 0);            // method: stored
    put16(blob_, 0);            // time
    put16(blob_, 0x21);         // date 1980-01-01
    put32(blob_, m.crc);
    put32(blob_, m.size);
    put32(blob_, m.size);
    put16(blob_, (uint16_t)m.file.size());
    put16(blob_, 0);
    blob_.insert(blob_.end(), m.file.begin(), m.file.end());
    blob_.insert(blob_.end(), img.begin(), img.end());
    members_.push_back(m);
}

bool NpzWriter::finish(const std::string& path, std::string& err)
{
    const uint32_t cd_off = (uint32_t)blob_.size();
    for (const Member& m : members_) {
        put32(blob_, 0x02014b50u);
        put16(blob_, 20); put16(blob_, 20); put16(blob_, 0); put16(blob_, 0); put16(blob_, 0); put16(blob_, 0x21);
        put32(blob_, m.crc); put32(blob_, m.size); put32(blob_, m.size);
        put16(blob_, (uint16_t)m.file.size());
        put16(blob_, 0); put16(blob_, 0); put16(blob_, 0); put16(blob_, 0);
        put32(blob_, 0);
        put32(blob_, m.offset);
        blob_.insert(blob_.end(), m.file.begin(), m.file.end());
    }
    const uint32_t cd_size = (uint32_t)blob_.size() - cd_off;
    put32(blob_, 0x06054b50u);
    put16(blob_, 0); put16(blob_, 0);
    put16(blob_, (uint16_t)members_.size()); put16(blob_, (uint16_t)members_.size());
    put32(blob_, cd_size); put32(blob_, cd_off);
    put16(blob_, 0);
    return write_all(path, blob_, err);
}

bool npz_read(const std::string& path, std::vector<std::pair<std::string, NpyArray>>& members, std::string& err)
{
    std::vector<uint8_t> buf;
    if (!read_all(path, buf, err)) return false;
    // end-of-central-directory record
    if (buf.size() < 22) { err = "not a zip archive: " + path; return false; }
    size_t eocd = std::string::npos;
    for (size_t i = buf.size() - 22 + 1; i-- > 0;) {
        if (get32(buf.data() + i) == 0x06054b50u) { eocd = i; break; }
        if (buf.size() - i > 22 + 65535) break;
    }
    if (eocd == std::string::npos) { err = "zip end record not found in " + path; return false; }
    const size_t n_ent = get16(buf.data() + eocd + 10);
    size_t pos = get32(buf.data() + eocd + 16);
    members.clear();
    for (size_t e = 0; e < n_ent; ++e) {
        if (pos + 46 > buf.size() || get32(buf.data() + pos) != 0x02014b50u) { err = "bad zip directory"; return false; }
        const uint16_t method = get16(buf.data() + pos + 10);
        const uint32_t csize = get32(buf.data() + pos + 20), usize = get32(buf.data() + pos + 24);
        const uint16_t nlen = get16(buf.data() + pos + 28), xlen = get16(buf.data() + pos + 30), clen = get16(buf.data() + pos + 32);
        const uint32_t lho = get32(buf.data() + pos + 42);
        std::string name(reinterpret_cast<const char*>(buf.data() + pos + 46), nlen);
        pos += 46 + (size_t)nlen + xlen + clen;
        if ((size_t)lho + 30 > buf.size() || get32(buf.data() + lho) != 0x04034b50u) { err = "bad zip local header"; return false; }
        const size_t data_off = (size_t)lho + 30 + get16(buf.data() + lho + 26) + get16(buf.data() + lho + 28);
        if (data_off + csize > buf.size()) { err = "truncated zip member " + name; return false; }
        std::vector<uint8_t> raw;
        if (method == 0) {
            raw.assign(buf.begin() + data_off, buf.begin() + data_off + csize);
        } else if (method == 8) {
            raw.resize(usize);
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) { err = "inflate init failed"; return false; }
            zs.next_in = buf.data() + data_off;
            zs.avail_in = csize;
            zs.next_out = raw.data();
            zs.avail_out = usize;
            const int zr = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (zr != Z_STREAM_END) { err = "inflate failed on " + name; return false; }
        } else {
            err = "unsupported zip compression in " + name;
            return false;
        }
        NpyArray arr;
        if (!npy_parse(raw.data(), raw.size(), arr, err)) { err += " (member " + name + ")"; return false; }
        if (name.size() > 4 && name.compare(name.size() - 4, 4, ".npy") == 0) name.resize(name.size() - 4);
        members.emplace_back(std::move(name), std::move(arr));
    }
    return true;
}

}  // namespace sz
