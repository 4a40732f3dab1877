/* tests/cpp/rust_shim_sequence.c -- the C-ABI call sequences of bindings/rust/src/gpu.rs, executed.
 *
 * The Rust shim cannot be compiled in this image (no rustc, crate bn 0.4.3 not vendored).  What it DOES to libzkgpu.so can be:
 * this program makes the same calls in the same order with the same argument conventions -- upload_dense, setup (zk_setup +
 * zk_crs_download), GpuProver::new (zk_qap_upload_dense + zk_crs_upload), prove_with_rs (zk_prove), verify (zk_crs_upload +
 * zk_verify), prove_stream (zk_host_alloc + zk_prove_submit_host / zk_prove_wait, two in flight), from_root_rep
 * (zk_qap_upload_sparse + zk_setup), from_root_rep_integers (zk_qap_upload_sparse_integers + zk_setup), MultiGpuProver (zk_comm_init + zk_mgpu_create / zk_mgpu_push_host / zk_mgpu_pop) -- and restates the shim's byte conversions in C (module bn_bytes: 32-byte big-endian <->
 * four little-endian words; bn's Fq2 packing, the 512-bit integer c1 * q + c0, by the same shift-subtract long division),
 * checked against values the Python twin computes (argv) and by round trips through the proof bytes.
 *
 *   rust_shim_sequence <simple.zk> <hex of 64-byte U512 packing of the G2 generator's x> <hex of y>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkgpu.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)
#define ZK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "FAILED %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, ctx ? zk_last_error(ctx) : ""); exit(1); } } while (0)

static zk_ctx* ctx = NULL;

/* ---- bn_bytes, restated ---- */
static const uint64_t Q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static void be32_to_words(const uint8_t* be, uint64_t w[4]) {
    for (int i = 0; i < 4; ++i) { w[i] = 0; for (int b = 0; b < 8; ++b) w[i] = (w[i] << 8) | be[(3 - i) * 8 + b]; }
}
static void words_to_be32(const uint64_t* w, uint8_t* out) {
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) out[(3 - i) * 8 + b] = (uint8_t)(w[i] >> (8 * (7 - b)));
}
static void u512_to_fq2(const uint8_t be[64], uint64_t c0[4], uint64_t c1[4]) {
    uint64_t rem[5] = {0, 0, 0, 0, 0}, quo[4] = {0, 0, 0, 0};
    for (int bit = 0; bit < 512; ++bit) {
        uint64_t inb = (be[bit / 8] >> (7 - bit % 8)) & 1;
        for (int k = 4; k >= 1; --k) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 63);
        rem[0] = (rem[0] << 1) | inb;
        int ge = rem[4] != 0;
        if (!ge) { ge = 1; for (int k = 3; k >= 0; --k) if (rem[k] != Q[k]) { ge = rem[k] > Q[k]; break; } }
        for (int k = 3; k >= 1; --k) quo[k] = (quo[k] << 1) | (quo[k - 1] >> 63);
        quo[0] <<= 1;
        if (ge) {
            uint64_t borrow = 0;
            for (int k = 0; k < 4; ++k) {
                uint64_t d1 = rem[k] - Q[k], b1 = rem[k] < Q[k];
                uint64_t d2 = d1 - borrow, b2 = d1 < borrow;
                rem[k] = d2; borrow = b1 | b2;
            }
            rem[4] -= borrow;
            quo[0] |= 1;
        }
    }
    memcpy(c0, rem, 32); memcpy(c1, quo, 32);
}
static void fq2_to_u512(const uint64_t* c0, const uint64_t* c1, uint8_t out[64]) {
    uint64_t acc[8] = {0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            unsigned __int128 t = (unsigned __int128)acc[i + j] + (unsigned __int128)c1[i] * Q[j] + carry;
            acc[i + j] = (uint64_t)t; carry = t >> 64;
        }
        acc[i + 4] = (uint64_t)carry;
    }
    unsigned __int128 carry = 0;
    for (int k = 0; k < 8; ++k) { unsigned __int128 t = (unsigned __int128)acc[k] + (k < 4 ? c0[k] : 0) + carry; acc[k] = (uint64_t)t; carry = t >> 64; }
    for (int k = 0; k < 8; ++k) for (int b = 0; b < 8; ++b) out[(7 - k) * 8 + b] = (uint8_t)(acc[k] >> (8 * (7 - b)));
}
static void unhex(const char* h, uint8_t* out, size_t n) {
    CHECK(strlen(h) == 2 * n);
    for (size_t i = 0; i < n; ++i) { unsigned v; sscanf(h + 2 * i, "%2x", &v); out[i] = (uint8_t)v; }
}
/* proof bytes -> bn's packing -> back, as proof_from_bytes / proof_to_bytes do for the G2 block */
static void g2_block_round_trip(const uint8_t in[129], uint8_t out[129]) {
    uint64_t w[16];
    be32_to_words(in + 1, w + 4); be32_to_words(in + 33, w); be32_to_words(in + 65, w + 12); be32_to_words(in + 97, w + 8);
    uint8_t bn[129];
    bn[0] = 4;
    fq2_to_u512(w, w + 4, bn + 1); fq2_to_u512(w + 8, w + 12, bn + 65);          /* g2_from_words: what bn's decoder is given */
    uint64_t v[16];
    u512_to_fq2(bn + 1, v, v + 4); u512_to_fq2(bn + 65, v + 8, v + 12);           /* g2_to_words: what bn's encoder gives back */
    CHECK(memcmp(v, w, sizeof w) == 0);
    out[0] = 4;
    words_to_be32(v + 4, out + 1); words_to_be32(v, out + 33); words_to_be32(v + 12, out + 65); words_to_be32(v + 8, out + 97);
}

static char* read_file(const char* path) {
    FILE* f = fopen(path, "rb");
    CHECK(f != NULL);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)n + 1);
    CHECK(fread(s, 1, (size_t)n, f) == (size_t)n);
    s[n] = 0; fclose(f);
    return s;
}
static void fr_small(uint64_t v, uint64_t w[4]) { w[0] = v; w[1] = w[2] = w[3] = 0; }

int main(int argc, char** argv) {
    CHECK(argc == 4);
    /* ---- byte conversions against the Python twin's values ---- */
    {
        /* G2 generator, words c0 | c1 */
        uint8_t xb[2][32], yb[2][32], want_x[64], want_y[64], got[64];
        unhex("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed", xb[0], 32);
        unhex("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2", xb[1], 32);
        unhex("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa", yb[0], 32);
        unhex("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b", yb[1], 32);
        unhex(argv[2], want_x, 64); unhex(argv[3], want_y, 64);
        uint64_t x0[4], x1[4], y0[4], y1[4], a[4], b[4];
        be32_to_words(xb[0], x0); be32_to_words(xb[1], x1); be32_to_words(yb[0], y0); be32_to_words(yb[1], y1);
        CHECK(x0[0] == 0x46debd5cd992f6edull && x0[3] == 0x1800deef121f1e76ull);
        fq2_to_u512(x0, x1, got); CHECK(memcmp(got, want_x, 64) == 0);
        fq2_to_u512(y0, y1, got); CHECK(memcmp(got, want_y, 64) == 0);
        u512_to_fq2(want_x, a, b); CHECK(memcmp(a, x0, 32) == 0 && memcmp(b, x1, 32) == 0);
        u512_to_fq2(want_y, a, b); CHECK(memcmp(a, y0, 32) == 0 && memcmp(b, y1, 32) == 0);
        /* extremes: c0 = c1 = q - 1 and zero */
        uint64_t qm1[4] = {Q[0] - 1, Q[1], Q[2], Q[3]}, z[4] = {0, 0, 0, 0};
        fq2_to_u512(qm1, qm1, got); u512_to_fq2(got, a, b); CHECK(memcmp(a, qm1, 32) == 0 && memcmp(b, qm1, 32) == 0);
        fq2_to_u512(z, z, got); u512_to_fq2(got, a, b); CHECK(memcmp(a, z, 32) == 0 && memcmp(b, z, 32) == 0);
        uint8_t be[32]; words_to_be32(x0, be); CHECK(memcmp(be, xb[0], 32) == 0);
        printf("ok byte_conversions\n");
    }

    /* ---- Ctx::new ---- */
    CHECK(zk_device_count() > 0);
    ZK(zk_ctx_create(0, &ctx));

    /* the dense QAP<CoefficientPoly<FrLocal>> of simple.zk, as the crate would hold it (here: parser + download) */
    char* code = read_file(argv[1]);
    zk_circuit* circ = NULL; char err[256];
    CHECK(zk_circuit_parse(code, &circ, err, sizeof err) == 0);
    size_t m, n, l, n_in;
    ZK(zk_circuit_dims(circ, &m, &n, &l, &n_in));
    CHECK(m == 6 && n == 2 && l == 2 && n_in == 3);
    zk_qap* q0 = NULL;
    ZK(zk_circuit_qap(ctx, circ, &q0));
    uint64_t *u = calloc(m * n * 4, 8), *v = calloc(m * n * 4, 8), *w = calloc(m * n * 4, 8), *t = calloc((n + 1) * 4, 8);
    ZK(zk_qap_download_dense(ctx, q0, u, v, w, t));
    zk_qap_free(q0);
    uint64_t inputs[12], weights[24];
    fr_small(3, inputs); fr_small(2, inputs + 4); fr_small(4, inputs + 8);
    ZK(zk_circuit_weights(circ, inputs, 3, weights, m));
    CHECK(weights[0] == 1 && weights[4] == 2 && weights[8] == 34 && weights[12] == 6 && weights[16] == 3 && weights[20] == 4);   /* circuit/mod.rs:759-766 */

    /* ---- gpu::setup: upload_dense, zk_setup(trapdoor), zk_crs_download, free ---- */
    zk_qap* q = NULL;
    ZK(zk_qap_upload_dense(ctx, u, v, w, t, m, n, l, &q));
    uint64_t td[20];
    for (int k = 0; k < 5; ++k) fr_small(0x1234567 + 977 * k, td + 4 * k);
    zk_crs* crs_dev = NULL;
    ZK(zk_setup(ctx, q, td, &crs_dev));
    uint64_t a1[8], b1[8], d1[8], b2[16], g2[16], d2[16];
    uint64_t *xi1 = calloc(8 * n, 8), *sg = calloc(8 * (l + 1), 8), *sd = calloc(8 * (m - l - 1), 8), *xt = calloc(8 * (n - 1) + 8, 8), *xi2 = calloc(16 * n, 8);
    zk_crs_out out = {a1, b1, d1, xi1, sg, sd, xt, b2, g2, d2, xi2};
    ZK(zk_crs_download(ctx, crs_dev, &out));
    zk_crs_free(crs_dev); zk_qap_free(q);
    printf("ok setup\n");

    /* ---- GpuProver::new: upload_dense + zk_crs_upload (host SigmaG1 / SigmaG2 -> device) ---- */
    ZK(zk_qap_upload_dense(ctx, u, v, w, t, m, n, l, &q));
    zk_crs_desc desc = {n, m, l, a1, b1, d1, xi1, sg, sd, xt, b2, g2, d2, xi2};
    zk_crs* crs = NULL;
    ZK(zk_crs_upload(ctx, &desc, &crs));

    /* ---- prove_with_rs: zk_prove, then the proof bytes through the bn packing and back ---- */
    uint64_t r[4], s[4];
    fr_small(0xabcdef12345ull, r); fr_small(0x777777ull, s);
    uint8_t proof[ZK_PROOF_BYTES], proof2[ZK_PROOF_BYTES];
    ZK(zk_prove(ctx, crs, q, weights, m, r, s, proof));
    CHECK(proof[0] == 4 && proof[65] == 4 && proof[194] == 4);
    memcpy(proof2, proof, ZK_PROOF_BYTES);
    g2_block_round_trip(proof + 65, proof2 + 65);
    CHECK(memcmp(proof, proof2, ZK_PROOF_BYTES) == 0);
    printf("ok prove\n");

    /* ---- GpuProver::from_root_rep_integers: the parser's root representation as it is (roots 1..n), zk_circuit_rows ->
     *      zk_qap_upload_sparse_integers + zk_setup with the same trapdoor: the same 259 bytes as the dense path above ---- */
    {
        size_t nnz[3];
        uint64_t* ptr[3]; uint32_t* gate[3]; uint64_t* val[3];
        for (int k = 0; k < 3; ++k) {
            ZK(zk_circuit_rows(circ, k, NULL, NULL, NULL, &nnz[k]));
            ptr[k] = calloc(m + 1, 8); gate[k] = calloc(nnz[k] + 1, 4); val[k] = calloc(4 * nnz[k] + 4, 8);
            ZK(zk_circuit_rows(circ, k, ptr[k], gate[k], val[k], &nnz[k]));
        }
        zk_qap_sparse_desc sd1 = {0, m, l, {ptr[0], gate[0], val[0]}, {ptr[1], gate[1], val[1]}, {ptr[2], gate[2], val[2]}};
        zk_qap* qs = NULL; zk_crs* cs = NULL;
        ZK(zk_qap_upload_sparse_integers(ctx, &sd1, n, &qs));
        ZK(zk_setup(ctx, qs, td, &cs));
        uint8_t ps[ZK_PROOF_BYTES];
        ZK(zk_prove(ctx, cs, qs, weights, m, r, s, ps));
        CHECK(memcmp(ps, proof, ZK_PROOF_BYTES) == 0);
        zk_crs_free(cs);
        /* ---- GpuProver::with_sigma_integers: the same rows with the HOST CRS of gpu::setup above (the reference's arrays only:
         *      zk_crs_upload; the library derives the Lagrange-basis points from [x^i] at the first proof) ---- */
        ZK(zk_crs_upload(ctx, &desc, &cs));
        memset(ps, 0, sizeof ps);
        ZK(zk_prove(ctx, cs, qs, weights, m, r, s, ps));
        CHECK(memcmp(ps, proof, ZK_PROOF_BYTES) == 0);
        CHECK(zk_qap_kind(qs) == 2);
        zk_crs_free(cs); zk_qap_free(qs);
        /* ---- GpuProver::from_root_rep_any: the same rows with the roots handed over as caller data (here 1, 2: the bytes above must
         *      come out again), zk_qap_upload_sparse_roots + the uploaded host CRS, and + zk_setup ---- */
        {
            uint64_t rts[8] = {1, 0, 0, 0, 2, 0, 0, 0};
            zk_qap* qa = NULL;
            ZK(zk_qap_upload_sparse_roots(ctx, &sd1, rts, n, &qa));
            CHECK(zk_qap_kind(qa) == 3);
            ZK(zk_crs_upload(ctx, &desc, &cs));
            memset(ps, 0, sizeof ps);
            ZK(zk_prove(ctx, cs, qa, weights, m, r, s, ps));
            CHECK(memcmp(ps, proof, ZK_PROOF_BYTES) == 0);
            zk_crs_free(cs);
            ZK(zk_setup(ctx, qa, td, &cs));
            memset(ps, 0, sizeof ps);
            ZK(zk_prove(ctx, cs, qa, weights, m, r, s, ps));
            CHECK(memcmp(ps, proof, ZK_PROOF_BYTES) == 0);
            zk_crs_free(cs); zk_qap_free(qa);
            printf("ok from_root_rep_any\n");
        }
        for (int k = 0; k < 3; ++k) { free(ptr[k]); free(gate[k]); free(val[k]); }
        printf("ok from_root_rep_integers\n");
    }

    /* ---- gpu::verify: zk_crs_upload of the host CRS, zk_verify (lib.rs:156-190: (2, 34) accepted, (2, 25) rejected) ---- */
    zk_crs* crs_v = NULL;
    ZK(zk_crs_upload(ctx, &desc, &crs_v));
    uint64_t vin[8]; int ok = -1;
    fr_small(2, vin); fr_small(34, vin + 4);
    ZK(zk_verify(ctx, crs_v, vin, 2, proof2, &ok)); CHECK(ok == 1);
    fr_small(25, vin + 4);
    ZK(zk_verify(ctx, crs_v, vin, 2, proof2, &ok)); CHECK(ok == 0);
    zk_crs_free(crs_v);
    printf("ok verify\n");

    /* ---- prove_stream: page-locked staging buffers, two tickets in flight ---- */
    {
        void* staging[2] = {NULL, NULL};
        int inflight[2], n_in_flight = 0, head = 0, done = 0;
        for (int k = 0; k < 5; ++k) {
            if (n_in_flight == 2) {
                uint8_t p[ZK_PROOF_BYTES];
                ZK(zk_prove_wait(ctx, inflight[head], p));
                CHECK(memcmp(p, proof, ZK_PROOF_BYTES) == 0);
                head ^= 1; --n_in_flight; ++done;
            }
            int slot = k % 2;
            if (!staging[slot]) ZK(zk_host_alloc(m * 32, &staging[slot]));
            memcpy(staging[slot], weights, m * 32);
            int ticket = -1;
            ZK(zk_prove_submit_host(ctx, crs, q, (const uint64_t*)staging[slot], m, r, s, &ticket));
            inflight[(head + n_in_flight) % 2] = ticket; ++n_in_flight;
        }
        while (n_in_flight) {
            uint8_t p[ZK_PROOF_BYTES];
            ZK(zk_prove_wait(ctx, inflight[head], p));
            CHECK(memcmp(p, proof, ZK_PROOF_BYTES) == 0);
            head ^= 1; --n_in_flight; ++done;
        }
        CHECK(done == 5);
        zk_host_free(staging[0]); zk_host_free(staging[1]);
        printf("ok prove_stream\n");
    }
    zk_crs_free(crs); zk_qap_free(q);

    /* ---- GpuProver::from_root_rep: sparse rows over the roots w^j (a 4-gate chain: t1 = x a1, t2 = x (t1 + a2), t3 = x (t2 + a3),
     *      y = 1 (t3 + a4); wires 1 x y t1 a1 t2 a2 t3 a3 a4), zk_qap_upload_sparse + zk_setup + zk_crs_download + prove + verify ---- */
    {
        const size_t sm = 10, sl = 2;
        /* CSR by wire: (gate, value 1) */
        uint64_t uptr[11] = {0, 1, 4, 4, 4, 4, 4, 4, 4, 4, 4};      uint32_t ug[4] = {3, 0, 1, 2};
        uint64_t vptr[11] = {0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7};      uint32_t vg[7] = {1, 0, 2, 1, 3, 2, 3};
        uint64_t wptr[11] = {0, 0, 0, 1, 2, 2, 3, 3, 4, 4, 4};      uint32_t wg[4] = {3, 0, 1, 2};
        uint64_t ones[7 * 4] = {0};
        for (int k = 0; k < 7; ++k) ones[4 * k] = 1;
        zk_qap_sparse_desc sd2 = {2, sm, sl, {uptr, ug, ones}, {vptr, vg, ones}, {wptr, wg, ones}};
        zk_qap* sq = NULL; zk_crs* scrs = NULL;
        ZK(zk_qap_upload_sparse(ctx, &sd2, &sq));
        ZK(zk_setup(ctx, sq, td, &scrs));
        /* witness: x = 3, a = (5, 7, 11, 13): t1 = 15, t2 = 66, t3 = 231, y = 244 */
        uint64_t wit[40] = {0};
        const uint64_t vals[10] = {1, 3, 244, 15, 5, 66, 7, 231, 11, 13};
        for (int k = 0; k < 10; ++k) wit[4 * k] = vals[k];
        uint8_t sp[ZK_PROOF_BYTES];
        ZK(zk_prove(ctx, scrs, sq, wit, sm, r, s, sp));
        fr_small(3, vin); fr_small(244, vin + 4);
        ZK(zk_verify(ctx, scrs, vin, 2, sp, &ok)); CHECK(ok == 1);
        fr_small(245, vin + 4);
        ZK(zk_verify(ctx, scrs, vin, 2, sp, &ok)); CHECK(ok == 0);
        printf("ok from_root_rep\n");
        /* ---- MultiGpuProver::{new, prove_stream}: zk_comm_init + zk_mgpu_create, then zk_mgpu_push_host two rounds ahead of
         *      zk_mgpu_pop.  One rank here (a round is `world` proofs, one per rank), so every proof must equal zk_prove's bytes:
         *      the exchange layout, the grouped inner products over this rank's points and zk_prove_combine are all on the path. ---- */
        {
            uint8_t id[ZK_COMM_ID_BYTES];
            zk_comm* comm = NULL; zk_mgpu* mg = NULL;
            void* hw = NULL;
            CHECK(zk_device_count() >= 1);
            ZK(zk_comm_unique_id(id));
            ZK(zk_comm_init(ctx, id, 0, 1, &comm));
            CHECK(zk_comm_rank(comm) == 0 && zk_comm_world(comm) == 1);
            ZK(zk_mgpu_create(ctx, comm, scrs, sq, &mg));
            ZK(zk_host_alloc(sizeof wit, &hw));
            memcpy(hw, wit, sizeof wit);
            uint8_t got[5][ZK_PROOF_BYTES];
            int pushed = 0, popped = 0;
            while (popped < 5) {
                while (pushed < 5 && pushed - popped < 3) {
                    int rc = zk_mgpu_push_host(mg, (const uint64_t*)hw, sm, (pushed & 1) ? s : r, (pushed & 1) ? r : s);
                    if (rc != 0) { fprintf(stderr, "zk_mgpu_push_host -> %d (%s)\n", rc, zk_mgpu_last_error(mg)); exit(1); }
                    ++pushed;
                }
                int rc = zk_mgpu_pop(mg, got[popped]);
                if (rc != 0) { fprintf(stderr, "zk_mgpu_pop -> %d (%s)\n", rc, zk_mgpu_last_error(mg)); exit(1); }
                ++popped;
            }
            uint8_t sp2[ZK_PROOF_BYTES];
            ZK(zk_prove(ctx, scrs, sq, wit, sm, s, r, sp2));
            for (int k = 0; k < 5; ++k) CHECK(memcmp(got[k], (k & 1) ? sp2 : sp, ZK_PROOF_BYTES) == 0);
            CHECK(zk_mgpu_pop(mg, got[0]) != 0);    /* nothing pushed */
            double t = 1.5;
            ZK(zk_comm_barrier(comm)); ZK(zk_comm_max_f64(comm, &t)); CHECK(t == 1.5);
            zk_mgpu_destroy(mg); zk_comm_destroy(comm); zk_host_free(hw);
            printf("ok multi_gpu_world_1\n");
        }
        zk_crs_free(scrs); zk_qap_free(sq);
    }
    zk_circuit_free(circ);
    zk_ctx_destroy(ctx);
    return 0;
}
