// pipeline.hpp -- device-resident QAP / CRS handles and the prove / setup pipelines.
#pragma once
#include "kernels.hpp"
#include "interp.hpp"

namespace zk {

// CSR with Fr values (Montgomery) on the device
struct DevCsr {
    DevBuf<uint32_t> ptr;   // rows+1
    DevBuf<uint32_t> idx;   // nnz column indices
    DevBuf<Fr> val;         // nnz
    size_t rows = 0, nnz = 0;
};

// tables of the integer-roots form (aproots.hip): (factorials up to 2n while they are built,) barycentric weights of R = {1..n} and of S = {n+1..2n-1},
// N(s) = t(s) on S, NTT image of the sequence 1/d (cyclic size 2^log_m >= 2n - 2); all Montgomery
struct ApTables {
    size_t n = 0;
    unsigned log_m = 0;
    DevBuf<Fr> fact, ifact, w, ws, ntab, bhat;
};

// tables of the arbitrary-roots form (arbroots.hip): the sub-product tree of the roots (t = its root node sits in zk_qap::dt)
struct ArbTables {
    std::shared_ptr<InterpTree> tree;
    std::vector<Fr> host_roots;    // Montgomery; t(x) of a trapdoor is a host product
};

}  // namespace zk

struct zk_qap {
    zk_ctx* ctx = nullptr;
    bool dense = false;
    int roots = 0;            // sparse form: 0 = roots of unity w^j (n = 2^log_n), 1 = the integers 1..n (any n; aproots.hip), 2 = caller's roots (arbroots.hip)
    std::shared_ptr<zk::ApTables> ap;
    std::shared_ptr<zk::ArbTables> arb;
    size_t n = 0, m = 0, input = 0;
    unsigned log_n = 0;
    // sparse form (roots w^j): by gate (prove: evaluation vectors) and by wire (setup: u_i(x))
    zk::DevCsr u_gate, v_gate;
    zk::DevCsr u_wire, v_wire, w_wire;
    // dense form: m x n coefficient matrices and t (n+1), Montgomery (arbitrary-roots form: t only)
    zk::DevBuf<zk::Fr> du, dv, dw, dt;
    size_t t_degree = 0;      // actual degree of t (dense)
    zk::DevBuf<zk::Fr> t_cinv; // 1 / leading coefficient of t (dense)
    bool t_is_zero = false;
    // dense form, large n: NTT image (size 2^tinv_log) of the power-series inverse of rev(t) mod x^K,
    // K = 2n-1-deg t quotient coefficients; built on first use (qap_ensure_tinv)
    zk::DevBuf<zk::Fr> t_rinv_ntt;
    unsigned tinv_log = 0;
};

struct zk_crs {
    zk_ctx* ctx = nullptr;
    size_t n = 0, m = 0, input = 0;
    zk::DevBuf<zk::G1A> alpha1, beta1, delta1;          // 1 each
    zk::DevBuf<zk::G1A> xi1, sum_gamma1, sum_delta1, xi_t1;
    zk::DevBuf<zk::G2A> beta2, gamma2, delta2, xi2;
    // integer-roots form (CRS made by zk_setup for such a QAP): the same CRS in the Lagrange basis of R = {1..n} and of
    // S = {n+1..2n-1}: [L_k(x)]_1, [L_k(x)]_2, [L^S_s(x) t(x)/delta]_1 -- public linear combinations of xi1 / xi2 / xi_t1
    bool ap = false;
    zk::DevBuf<zk::G1A> lag1, lagS_t1;
    zk::DevBuf<zk::G2A> lag2;
    // bit-reversed copies for the roots-of-unity pipeline (built on first use); xi_t1_br has n
    // entries, the last one (coefficient n-1, never used by the reference) is infinity
    zk::DevBuf<zk::G1A> xi1_br, xi_t1_br;
    zk::DevBuf<zk::G2A> xi2_br;
    unsigned br_log_n = 0;
    bool has_br = false;
    // fixed-base window tables T[w][i] = 2^(c w) P_i for the four base sets prove() uses
    // (built on first use for the order -- natural or bit-reversed -- the QAP kind needs)
    zk::MsmTable<zk::Fq> t_xi1, t_hb1;   // t_hb1: bases xi_t | xi | sum_delta (H, r*B1, s*A and L in one product); sum_delta starts at point off_l
    size_t off_l = 0;
    zk::MsmTable<zk::Fq2> t_xi2;
    // per-rank tables of the multi-GPU scalar exchange: only the points [g c, (g+1) c) of every product that rank g of `world`
    // multiplies (1 / world of the table memory and of the time to build them); one (rank, world) at a time
    struct RankTables {
        int rank = -1, world = 0, kind = -1;
        long c_opt = -1;
        zk::MsmTable<zk::Fq> t_xi1, t_hb1;   // t_hb1: the rank's points of xi_t | xi (padded to ch) | its points of sum_delta (from point off_l = ch on)
        size_t off_l = 0;
        zk::MsmTable<zk::Fq2> t_xi2;
    } rank_tabs;
    int tables_kind = -1;    // -1 none, 0 natural order, 1 bit-reversed, 2 Lagrange-basis points (integer roots)
    // 4-bit fixed-base tables FT[w][d] = d * 16^w * P (64 x 16 entries) for the single CRS points
    // that prove() multiplies by r, s and r*s
    zk::DevBuf<zk::G1A> ft_alpha1, ft_beta1, ft_delta1;
    zk::DevBuf<zk::G2A> ft_beta2, ft_delta2;
    bool has_ft = false;
    long tables_c = -1;      // the msm_window_bits option the tables were built for
};

namespace zk {

zk_qap* qap_upload_sparse(zk_ctx*, const zk_qap_sparse_desc&);
zk_qap* qap_upload_rows(zk_ctx*, const zk_qap_sparse_desc&, size_t n);                  // rows over n gates, domain not set
zk_qap* qap_upload_sparse_integers(zk_ctx*, const zk_qap_sparse_desc&, size_t n);       // aproots.hip
zk_qap* qap_upload_sparse_roots(zk_ctx*, const zk_qap_sparse_desc&, const uint64_t* roots, size_t n);   // arbroots.hip
void arb_download_roots(zk_ctx*, const zk_qap&, uint64_t* out);
void arb_setup_lagrange(zk_ctx*, const zk_qap&, const uint64_t trapdoor[20], Fr* d_L, int* d_flag);
Fr arb_t_at_x(const zk_qap&, const uint64_t trapdoor[20]);
size_t arb_work_elems(const zk_qap&);
void arb_coefficients(zk_ctx*, const zk_qap&, const Fr* vals, Fr* work, Fr* uc, Fr* vc);
void ap_setup_lagrange(zk_ctx*, const zk_qap&, const uint64_t trapdoor[20], Fr* d_L, Fr* d_LS, int* d_flag);
Fr ap_t_at_x(const zk_qap&, const uint64_t trapdoor[20]);
void ap_quotient_values(zk_ctx*, const zk_qap&, const Fr* ue, const Fr* ve, Fr* work, Fr* hb_can, size_t count = 1, size_t hb_stride = 0);
zk_qap* qap_upload_dense(zk_ctx*, const uint64_t* u, const uint64_t* v, const uint64_t* w, const uint64_t* t, size_t m, size_t n, size_t input);
void qap_free(zk_qap*);
void qap_weighted_sum(zk_ctx*, const zk_qap&, const uint64_t* weights, size_t m, int which, uint64_t* out);   // qap.hip
void qap_save(zk_ctx*, const zk_qap&, const char* path);      // serialize.hip
zk_qap* qap_load(zk_ctx*, const char* path);
void proof_save(const uint8_t proof[ZK_PROOF_BYTES], const char* path);
void proof_load(const char* path, uint8_t proof[ZK_PROOF_BYTES]);

zk_crs* crs_upload(zk_ctx*, const zk_crs_desc&);
zk_crs* crs_setup(zk_ctx*, const zk_qap&, const uint64_t trapdoor[20]);
void crs_dims(const zk_crs&, size_t* n, size_t* m, size_t* input);
void crs_download(zk_ctx*, const zk_crs&, const zk_crs_out&);
void crs_save(zk_ctx*, const zk_crs&, const char* path);     // serialize.hip
zk_crs* crs_load(zk_ctx*, const char* path);
void crs_free(zk_crs*);
void crs_ensure_brev(zk_ctx*, zk_crs&, unsigned log_n);
void crs_ensure_tables(zk_ctx*, zk_crs&, bool brev, unsigned log_n, bool lagrange = false);
void crs_ensure_rank_tables(zk_ctx*, zk_crs&, bool brev, unsigned log_n, bool lagrange, int rank, int world, size_t cl, size_t cn, size_t ch);
void crs_download_lagrange(zk_ctx*, const zk_crs&, uint64_t* lag1, uint64_t* lagS_t1, uint64_t* lag2);
void crs_attach_lagrange(zk_ctx*, zk_crs&, const uint64_t* lag1, const uint64_t* lagS_t1, const uint64_t* lag2);
void crs_ensure_fixed_tables(zk_ctx*, zk_crs&);
size_t basis_max_n();                                              // the largest n crs_lagrange_from_powers takes
void arb_attach_integer_roots(zk_ctx*, zk_qap&);                    // arbroots.hip: the tree of the roots 1..n for an integer-roots QAP (see prove.hip)
void crs_lagrange_from_powers_tree(zk_ctx*, zk_crs&, const zk_qap&);   // gbasis.hip: the same arrays by the transposed interpolation tree, O(n log^2 n) point operations
void crs_lagrange_from_powers(zk_ctx*, zk_crs&, const zk_qap&);   // basis.hip: the Lagrange-basis points of an integer-roots QAP from the powers (once per CRS)

void prove_host(zk_ctx*, const zk_crs&, const zk_qap&, const uint64_t* weights, size_t m, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out);
// rank/world select the owned Pippenger windows.  With d_partial_out == nullptr the proof is
// finished locally (world must be 1); otherwise the five partial sums are written there.
void prove_dev(zk_ctx*, const zk_crs&, const zk_qap&, const Fr* d_weights, size_t m, const uint64_t* r, const uint64_t* s,
               uint8_t* proof_out, int rank, int world, void* d_partial_out);
int prove_submit(zk_ctx*, const zk_crs&, const zk_qap&, const Fr* d_weights, size_t m, const uint64_t* r, const uint64_t* s,
                 int rank, int world, void* d_partial_out, Fr* const* xout = nullptr);
// scalar exchange: element counts of the four exchange arrays (L | V | U | H,k), and the inner products of `sets`
// proofs over this rank's points from the chunks an all-to-all delivered
void prove_exchange_elems(const zk_qap&, int world, size_t out[4]);
int prove_msm_submit(zk_ctx*, const zk_crs&, const zk_qap&, int sets, int rank, int world,
                     const Fr* d_l, const Fr* d_vc, const Fr* d_uc, const Fr* d_hb, void* d_partials_out);
int prove_submit_host(zk_ctx*, const zk_crs&, const zk_qap&, const uint64_t* weights, size_t m, const uint64_t* r, const uint64_t* s,
                      int world = 1, Fr* const* xout = nullptr);
void prove_wait(zk_ctx*, int ticket, uint8_t* proof_out);
// stream-ordered use of tickets (comm.hip): the stream a ticket completes on, and release of its slot WITHOUT waiting (h_flag_pinned,
// may be null: where the witness range flag of a scalars ticket is copied once the ticket is complete)
void ctx_reserve_cus(zk_ctx*, int per_xcd);   // capi.hip: inner-product streams masked to leave compute units to the collectives
hipStream_t prove_ticket_stream(zk_ctx*, int ticket);
void prove_release(zk_ctx*, int ticket, int* h_flag_pinned);
// prove_combine without the host synchronisation: everything on `st`, the proof lands in h_proof_pinned
void prove_combine_on(zk_ctx*, const zk_crs&, const void* d_partials, int world, const uint64_t r[4], const uint64_t s[4], hipStream_t st, uint8_t* h_proof_pinned);
int prove_batch_submit(zk_ctx*, const zk_crs&, const zk_qap&, int count, const void* const* d_weights, const size_t* m,
                       const uint64_t* r, const uint64_t* s);
void prove_batch_wait(zk_ctx*, int ticket, int count, uint8_t* proofs_out);
void prove_combine(zk_ctx*, const zk_crs&, const void* d_partials, int world, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out);

}  // namespace zk
