// serialize.hip -- on-disk containers for the CRS, the QAP and proofs.
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
// A CRS that zk_setup made for an integer-roots QAP (aproots.hip) also carries the same CRS in the Lagrange bases; its file is
// "ZKCRSv2\0" with the same header and, behind the v1 payload,
//              lag1 (n x 64) | lagS_t1 ((n-1) x 64) | lag2 (n x 128)
// so that a reloaded CRS proves for that QAP form again.  zk_crs_load reads both versions.
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
constexpr char MAGIC2[8] = {'Z', 'K', 'C', 'R', 'S', 'v', '2', '\0'};

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
    const size_t n = crs.n, extra = crs.ap ? 8 * n + 8 * (n - 1) + 16 * n : 0;
    std::vector<uint64_t> buf(lay.words() + extra);
    uint64_t* b = buf.data();
    if (crs.ap) crs_download_lagrange(ctx, crs, b + lay.words(), b + lay.words() + 8 * n, b + lay.words() + 8 * n + 8 * (n - 1));
    zk_crs_out out{b + lay.off[0], b + lay.off[1], b + lay.off[2], b + lay.off[3], b + lay.off[4], b + lay.off[5], b + lay.off[6],
                   b + lay.off[7], b + lay.off[8], b + lay.off[9], b + lay.off[10]};
    crs_download(ctx, crs, out);
    uint64_t head[5];
    std::memcpy(head, crs.ap ? MAGIC2 : MAGIC, 8);
    head[1] = crs.n; head[2] = crs.m; head[3] = crs.input;
    head[4] = fnv1a(reinterpret_cast<const uint8_t*>(b), buf.size() * 8);
    File f(std::fopen(path, "wb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("crs_save: cannot open ") + path);
    bool ok = std::fwrite(head, 8, 5, f.f) == 5 && std::fwrite(b, 8, buf.size(), f.f) == buf.size();
    ok = ok && std::fflush(f.f) == 0;
    ZK_REQUIRE(ok, ZK_ERR_IO, std::string("crs_save: short write to ") + path);
}

// bytes left in the file behind the current position (the loaders size their buffers from header fields: a 72-byte file must not
// be able to ask for tens of GB)
static uint64_t bytes_left(std::FILE* f) {
    const long at = std::ftell(f);
    if (at < 0 || std::fseek(f, 0, SEEK_END) != 0) return 0;
    const long end = std::ftell(f);
    (void)std::fseek(f, at, SEEK_SET);
    return end > at ? (uint64_t)(end - at) : 0;
}

zk_crs* crs_load(zk_ctx* ctx, const char* path) {
    File f(std::fopen(path, "rb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("crs_load: cannot open ") + path);
    uint64_t head[5];
    ZK_REQUIRE(std::fread(head, 8, 5, f.f) == 5 && (!std::memcmp(head, MAGIC, 8) || !std::memcmp(head, MAGIC2, 8)), ZK_ERR_IO, "crs_load: not a ZKCRSv1 / ZKCRSv2 file");
    const bool v2 = !std::memcmp(head, MAGIC2, 8);
    const size_t n = head[1], m = head[2], input = head[3];
    ZK_REQUIRE(n >= 1 && n <= ((size_t)1 << 26) && m >= input + 1 && m <= ((size_t)1 << 28), ZK_ERR_IO, "crs_load: implausible dimensions in the header");
    Layout lay(n, m, input);
    const size_t crs_words = lay.words() + (v2 ? 8 * n + 8 * (n - 1) + 16 * n : 0);
    ZK_REQUIRE(bytes_left(f.f) >= (uint64_t)crs_words * 8, ZK_ERR_IO, "crs_load: file is truncated");
    std::vector<uint64_t> buf(crs_words);
    ZK_REQUIRE(std::fread(buf.data(), 8, buf.size(), f.f) == buf.size(), ZK_ERR_IO, "crs_load: file is truncated");
    uint8_t extra;
    ZK_REQUIRE(std::fread(&extra, 1, 1, f.f) == 0, ZK_ERR_IO, "crs_load: trailing bytes after the payload");
    ZK_REQUIRE(fnv1a(reinterpret_cast<const uint8_t*>(buf.data()), buf.size() * 8) == head[4], ZK_ERR_IO, "crs_load: checksum mismatch");
    const uint64_t* b = buf.data();
    zk_crs_desc d{n, m, input, b + lay.off[0], b + lay.off[1], b + lay.off[2], b + lay.off[3], b + lay.off[4], b + lay.off[5], b + lay.off[6],
                  b + lay.off[7], b + lay.off[8], b + lay.off[9], b + lay.off[10]};
    std::unique_ptr<zk_crs, void (*)(zk_crs*)> c(crs_upload(ctx, d), crs_free);
    if (v2) crs_attach_lagrange(ctx, *c, b + lay.words(), b + lay.words() + 8 * n, b + lay.words() + 8 * n + 8 * (n - 1));
    return c.release();
}

// ---- QAP container ("ZKQAPv1") and proof file ("ZKPRFv1") ------------------------------------------------------------
// The reference cannot store a QAP either (QAP<P> has private fields and no serialisation, groth16/mod.rs:60-67); the container
// holds what zk_qap_upload_sparse / zk_qap_upload_dense take, as canonical little-endian integers:
//   offset 0   "ZKQAPv1\0"
//          8   kind (0 = sparse rows over the roots w^j, 1 = dense coefficient matrices, 2 = sparse rows over the
//              integer roots 1..n, 3 = sparse rows over the caller's roots), n_or_log_n (log_n for kind 0, n otherwise), m, input   4 x u64
//         40   nnz(u), nnz(v), nnz(w)  (0 for the dense kind)                                                    3 x u64
//         64   FNV-1a 64 of the payload
//         72   payload  sparse: for u, v, w: ptr[m+1] (u64) | gate[nnz] (u32, padded to 8 bytes) | val[nnz] (32 B); kind 3: then the n roots (32 B)
//                       dense : u, v, w (m n x 32 B each), t ((n+1) x 32 B)
// Loading goes through the upload entry points, so every value is range-checked on the GPU.
namespace {
constexpr char QMAGIC[8] = {'Z', 'K', 'Q', 'A', 'P', 'v', '1', '\0'};
constexpr char PMAGIC[8] = {'Z', 'K', 'P', 'R', 'F', 'v', '1', '\0'};

void csr_to_host(zk_ctx* ctx, const DevCsr& c, size_t rows, std::vector<uint64_t>& ptr, std::vector<uint32_t>& gate, std::vector<uint64_t>& val) {
    std::vector<uint32_t> p32(rows + 1);
    ZK_HIP(hipMemcpy(p32.data(), c.ptr.p, (rows + 1) * 4, hipMemcpyDeviceToHost));
    ptr.assign(p32.begin(), p32.end());
    const size_t nnz = p32[rows];
    gate.assign(nnz + (nnz & 1), 0);
    val.assign(nnz * 4, 0);
    if (!nnz) return;
    ZK_HIP(hipMemcpy(gate.data(), c.idx.p, nnz * 4, hipMemcpyDeviceToHost));
    DevBuf<Fr> tmp(nnz);
    fr_from_mont(ctx, c.val.p, tmp.p, nnz);
    ZK_HIP(hipMemcpyAsync(val.data(), tmp.p, nnz * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}
}  // namespace

void qap_save(zk_ctx* ctx, const zk_qap& q, const char* path) {
    std::vector<uint64_t> payload;
    uint64_t head[9] = {0};
    std::memcpy(head, QMAGIC, 8);
    head[1] = q.dense ? 1 : (q.roots == 2 ? 3 : q.roots ? 2 : 0); head[2] = (q.dense || q.roots) ? q.n : q.log_n; head[3] = q.m; head[4] = q.input;
    if (!q.dense) {
        const DevCsr* rows[3] = {&q.u_wire, &q.v_wire, &q.w_wire};
        for (int k = 0; k < 3; ++k) {
            std::vector<uint64_t> ptr, val;
            std::vector<uint32_t> gate;
            csr_to_host(ctx, *rows[k], q.m, ptr, gate, val);
            head[5 + k] = ptr[q.m];
            payload.insert(payload.end(), ptr.begin(), ptr.end());
            const size_t at = payload.size();
            payload.resize(at + gate.size() / 2);
            std::memcpy(payload.data() + at, gate.data(), gate.size() * 4);
            payload.insert(payload.end(), val.begin(), val.end());
        }
        if (q.roots == 2) {
            const size_t at = payload.size();
            payload.resize(at + 4 * q.n);
            arb_download_roots(ctx, q, payload.data() + at);
        }
    } else {
        const size_t mn = q.m * q.n;
        payload.resize((3 * mn + q.n + 1) * 4);
        DevBuf<Fr> tmp(std::max(mn, q.n + 1));
        const Fr* src[4] = {q.du.p, q.dv.p, q.dw.p, q.dt.p};
        for (int k = 0; k < 4; ++k) {
            const size_t cnt = k < 3 ? mn : q.n + 1;
            fr_from_mont(ctx, src[k], tmp.p, cnt);
            ZK_HIP(hipMemcpyAsync(payload.data() + (size_t)k * mn * 4, tmp.p, cnt * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
            ZK_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    head[8] = fnv1a(reinterpret_cast<const uint8_t*>(payload.data()), payload.size() * 8);
    File f(std::fopen(path, "wb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("qap_save: cannot open ") + path);
    bool ok = std::fwrite(head, 8, 9, f.f) == 9 && std::fwrite(payload.data(), 8, payload.size(), f.f) == payload.size() && std::fflush(f.f) == 0;
    ZK_REQUIRE(ok, ZK_ERR_IO, std::string("qap_save: short write to ") + path);
}

zk_qap* qap_load(zk_ctx* ctx, const char* path) {
    File f(std::fopen(path, "rb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("qap_load: cannot open ") + path);
    uint64_t head[9];
    ZK_REQUIRE(std::fread(head, 8, 9, f.f) == 9 && !std::memcmp(head, QMAGIC, 8), ZK_ERR_IO, "qap_load: not a ZKQAPv1 file");
    const uint64_t kind = head[1], m = head[3], input = head[4];
    ZK_REQUIRE(kind <= 3 && m >= 1 && m <= ((uint64_t)1 << 31) && input < m, ZK_ERR_IO, "qap_load: implausible header");
    size_t words;
    if (kind != 1) {
        ZK_REQUIRE(head[2] <= (kind == 0 ? 26 : ((uint64_t)1 << 23)) && head[5] <= ((uint64_t)1 << 32) && head[6] <= ((uint64_t)1 << 32) && head[7] <= ((uint64_t)1 << 32), ZK_ERR_IO, "qap_load: implausible header");
        words = 0;
        for (int k = 0; k < 3; ++k) words += (m + 1) + (head[5 + k] + 1) / 2 + head[5 + k] * 4;
        if (kind == 3) words += 4 * head[2];
    } else {
        ZK_REQUIRE(head[2] >= 1 && head[2] <= ((uint64_t)1 << 22) && m * head[2] <= ((uint64_t)1 << 33), ZK_ERR_IO, "qap_load: implausible header");
        words = (3 * m * head[2] + head[2] + 1) * 4;
    }
    ZK_REQUIRE(bytes_left(f.f) >= (uint64_t)words * 8, ZK_ERR_IO, "qap_load: file is truncated");
    std::vector<uint64_t> buf(words);
    ZK_REQUIRE(std::fread(buf.data(), 8, words, f.f) == words, ZK_ERR_IO, "qap_load: file is truncated");
    uint8_t extra;
    ZK_REQUIRE(std::fread(&extra, 1, 1, f.f) == 0, ZK_ERR_IO, "qap_load: trailing bytes after the payload");
    ZK_REQUIRE(fnv1a(reinterpret_cast<const uint8_t*>(buf.data()), words * 8) == head[8], ZK_ERR_IO, "qap_load: checksum mismatch");
    if (kind == 1) {
        const size_t mn = m * head[2];
        return qap_upload_dense(ctx, buf.data(), buf.data() + mn * 4, buf.data() + 2 * mn * 4, buf.data() + 3 * mn * 4, m, head[2], input);
    }
    zk_qap_sparse_desc d{};
    d.log_n = (unsigned)head[2]; d.m = m; d.input = input;
    zk_sparse_rows* rows[3] = {&d.u, &d.v, &d.w};
    const uint64_t* at = buf.data();
    for (int k = 0; k < 3; ++k) {
        const size_t nnz = head[5 + k];
        rows[k]->ptr = at; at += m + 1;
        ZK_REQUIRE(rows[k]->ptr[m] == nnz, ZK_ERR_IO, "qap_load: row offsets disagree with the header");
        rows[k]->gate = reinterpret_cast<const uint32_t*>(at); at += (nnz + 1) / 2;
        rows[k]->val = at; at += nnz * 4;
    }
    if (kind == 3) return qap_upload_sparse_roots(ctx, d, at, head[2]);
    return kind == 2 ? qap_upload_sparse_integers(ctx, d, head[2]) : qap_upload_sparse(ctx, d);
}

// A proof on disk: magic (the version lives in it), the 259 canonical bytes, FNV-1a 64 of them.  The in-memory encoding of the
// ABI stays the bare 259 bytes -- the reference's Proof has no encoding at all (mod.rs:124-128).
void proof_save(const uint8_t proof[ZK_PROOF_BYTES], const char* path) {
    File f(std::fopen(path, "wb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("proof_save: cannot open ") + path);
    const uint64_t sum = fnv1a(proof, ZK_PROOF_BYTES);
    bool ok = std::fwrite(PMAGIC, 1, 8, f.f) == 8 && std::fwrite(proof, 1, ZK_PROOF_BYTES, f.f) == ZK_PROOF_BYTES && std::fwrite(&sum, 8, 1, f.f) == 1 && std::fflush(f.f) == 0;
    ZK_REQUIRE(ok, ZK_ERR_IO, std::string("proof_save: short write to ") + path);
}
void proof_load(const char* path, uint8_t proof[ZK_PROOF_BYTES]) {
    File f(std::fopen(path, "rb"));
    ZK_REQUIRE(f.f, ZK_ERR_IO, std::string("proof_load: cannot open ") + path);
    char magic[8];
    uint64_t sum;
    uint8_t extra;
    ZK_REQUIRE(std::fread(magic, 1, 8, f.f) == 8 && !std::memcmp(magic, PMAGIC, 8), ZK_ERR_IO, "proof_load: not a ZKPRFv1 file");
    ZK_REQUIRE(std::fread(proof, 1, ZK_PROOF_BYTES, f.f) == ZK_PROOF_BYTES && std::fread(&sum, 8, 1, f.f) == 1, ZK_ERR_IO, "proof_load: file is truncated");
    ZK_REQUIRE(std::fread(&extra, 1, 1, f.f) == 0, ZK_ERR_IO, "proof_load: trailing bytes");
    ZK_REQUIRE(sum == fnv1a(proof, ZK_PROOF_BYTES), ZK_ERR_IO, "proof_load: checksum mismatch");
    ZK_REQUIRE((proof[0] == 0 || proof[0] == 4) && (proof[65] == 0 || proof[65] == 4) && (proof[194] == 0 || proof[194] == 4), ZK_ERR_IO, "proof_load: unknown point tag");
}

}  // namespace zk
