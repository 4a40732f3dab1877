// serialize.hip -- on-disk container for the CRS.
//
// The reference keeps (SigmaG1, SigmaG2) (/root/reference/src/groth16/mod.rs:105-121) in memory only
// and has no serialisation; re-running groth16::setup (mod.rs:134-197) draws a new
// trapdoor, so a CRS that is not written down cannot be reproduced.  The container stores exactly the
// arrays of zk_crs_desc as canonical little-endian integers (the ABI's own element convention), so a
// file is valid input for zk_crs_upload on any build:
//
//   offset 0   "ZKCRSv1\0"                                   8 bytes
//          8   n, m, input                                   3 x u64 LE
//         32   FNV-1a 64 of the payload                      u64 LE
//         40   payload: alpha1 beta1 delta1 (64 B each) | xi1 (n x 64) | sum_gamma1 ((input+1) x 64) |
//              sum_delta1 ((m-input-1) x 64) | xi_t1 ((n-1) x 64) | beta2 gamma2 delta2 (128 B each) | xi2 (n x 128)
//
// Loading range-checks every coordinate on the GPU (zk_crs_upload) and rejects truncated or altered files.
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#include "pipeline.hpp"

namespace zk {

namespace {
constexpr char MAGIC[8] = {'Z', 'K', 'C', 'R', 'S', 'v', '1', '\0'};

struct Layout {
    size_t n, m, input;
    size_t off[12];   // word offsets of the 11 arrays + total
    Layout(size_t n_, size_t m_, size_t l_) : n(n_), m(m_), input(l_) {
        const size_t cnt[11] = {8, 8, 8, 8 * n, 8 * (input + 1), 8 * (m - input - 1), 8 * (n - 1), 16, 16, 16, 16 * n};
        off[0] = 0;
        for (int k = 0; k < 11; ++k) off[k + 1] = off[k] + cnt[k];
    }
    size_t words() const { return off[11]; }
};

uint64_t fnv1a(const uint8_t* p, size_t len) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < len; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

struct File {
    FILE* f;
    explicit File(FILE* f_) : f(f_) {}
    ~File() { if (f) std::fclose(f); }
};
}  // namespace

void crs_save(zk_ctx* ctx, const zk_crs& crs, const char* path) {
    ZK_REQUIRE(crs.m >= crs.input + 1 && crs.n >= 1, ZK_ERR_ARG, "crs_save: inconsistent CRS dimensions");
    Layout lay(crs.n, crs.m, crs.input);
    std::vector<uint64_t> buf(lay.words());
    uint64_t* b = buf.data();
    zk_crs_out out{b + lay.off[0], b + lay.off[1], b + lay.off[2], b + lay.off[3], b + lay.off[4], b + lay.off[5], b + lay.off[6],
                   b + lay.off[7], b + lay.off[8], b + lay.off[9], b + lay.off[10]};
    crs_download(ctx, crs, out);
    uint64_t head[5];
    std::memcpy(head, MAGIC, 8);
    head[1] = crs.n; head[2] = crs.m; head[3] = crs.input;
    head[4] = fnv1a(reinterpret_cast<const uint8_t*>(b), buf.size() * 8);
    File f(std::fopen(path, "wb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("crs_save: cannot open ") + path);
    bool ok = std::fwrite(head, 8, 5, f.f) == 5 && std::fwrite(b, 8, buf.size(), f.f) == buf.size();
    ok = ok && std::fflush(f.f) == 0;
    ZK_REQUIRE(ok, ZK_ERR_IO, std::string("crs_save: short write to ") + path);
}

zk_crs* crs_load(zk_ctx* ctx, const char* path) {
    File f(std::fopen(path, "rb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("crs_load: cannot open ") + path);
    uint64_t head[5];
    ZK_REQUIRE(std::fread(head, 8, 5, f.f) == 5 && !std::memcmp(head, MAGIC, 8), ZK_ERR_IO, "crs_load: not a ZKCRSv1 file");
    const size_t n = head[1], m = head[2], input = head[3];
    ZK_REQUIRE(n >= 1 && n <= ((size_t)1 << 26) && m >= input + 1 && m <= ((size_t)1 << 28), ZK_ERR_IO, "crs_load: implausible dimensions in the header");
    Layout lay(n, m, input);
    std::vector<uint64_t> buf(lay.words());
    ZK_REQUIRE(std::fread(buf.data(), 8, buf.size(), f.f) == buf.size(), ZK_ERR_IO, "crs_load: file is truncated");
    uint8_t extra;
    ZK_REQUIRE(std::fread(&extra, 1, 1, f.f) == 0, ZK_ERR_IO, "crs_load: trailing bytes after the payload");
    ZK_REQUIRE(fnv1a(reinterpret_cast<const uint8_t*>(buf.data()), buf.size() * 8) == head[4], ZK_ERR_IO, "crs_load: checksum mismatch");
    const uint64_t* b = buf.data();
    zk_crs_desc d{n, m, input, b + lay.off[0], b + lay.off[1], b + lay.off[2], b + lay.off[3], b + lay.off[4], b + lay.off[5], b + lay.off[6],
                  b + lay.off[7], b + lay.off[8], b + lay.off[9], b + lay.off[10]};
    return crs_upload(ctx, d);
}

}  // namespace zk
