/* zkgpu.h -- C ABI of the MI355X-native Groth16 prover (drop-in boundary).
 *
 * The reference (republicprotocol/zksnark-rs, crate `zksnark` 0.0.2) is pure Rust with no FFI of
 * its own; this header IS the seam a maintainer binds from
 * `src/groth16/gpu.rs` (binding shown in INTEGRATION.md).  Every entry point cites the
 * reference interface it replaces (file:line into the reference tree).
 *
 * Conventions
 *   - Return value: 0 (ZK_OK) or a negative zk_status; nothing aborts or throws across the ABI.
 *     The reference panics instead (fr.rs:54,69; field/mod.rs:440); the Rust shim turns a
 *     non-zero status back into the same panic.
 *   - Fr / Fq element: uint64_t[4], little-endian limbs, CANONICAL integer < modulus
 *     (not Montgomery).  Conversions to the device's Montgomery form happen on the GPU.
 *   - G1 affine point: uint64_t[8]  = x[4] | y[4];                 infinity = all zero.
 *   - G2 affine point: uint64_t[16] = x.c0 | x.c1 | y.c0 | y.c1;   infinity = all zero.
 *     (Fq2 = Fq[i]/(i^2+1), element c0 + c1*i.)
 *   - Proof bytes (build-defined canonical encoding): 259 bytes
 *       A (G1, 65 B) | B (G2, 129 B) | C (G1, 65 B)
 *       G1: 0x04 | x | y                          (32-byte big-endian each)
 *       G2: 0x04 | x.c1 | x.c0 | y.c1 | y.c0      (EIP-197 order)
 *       infinity: 0x00 followed by zero bytes (same total length).
 *   - A zk_ctx is bound to one HIP device and must be used from one thread at a time.
 *   - All `const uint64_t*` inputs are HOST pointers unless the parameter name starts with d_.
 */
#ifndef ZKGPU_H
#define ZKGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;
typedef struct zk_crs zk_crs;   /* device-resident SigmaG1 + SigmaG2 (groth16/mod.rs:105-121) */
typedef struct zk_qap zk_qap;   /* device-resident QAP (groth16/mod.rs:60-67) */

typedef enum {
    ZK_OK = 0,
    ZK_ERR_ARG = -1,          /* null pointer / inconsistent sizes */
    ZK_ERR_HIP = -2,          /* HIP runtime error (see zk_last_error) */
    ZK_ERR_NO_DEVICE = -3,    /* no gfx950 device visible: the product path has NO CPU fallback */
    ZK_ERR_SIZE = -4,         /* size outside the supported range */
    ZK_ERR_DIV_BY_ZERO = -5,  /* "Dividend must be non-zero" (field/mod.rs:440) / Fr inverse of 0 (fr.rs:54,69) */
    ZK_ERR_RANGE = -6,        /* an Fr/Fq input is >= its modulus */
    ZK_ERR_UNSUPPORTED = -7,
    ZK_ERR_IO = -8,           /* file missing, truncated, altered or not in the expected format */
    ZK_ERR_COMM = -9          /* a collective failed (RCCL error or the caller's transport returned non-zero) */
} zk_status;

#define ZK_PROOF_BYTES 259
#define ZK_MAX_IN_FLIGHT 4   /* proofs one context can have submitted and not yet waited for */
#define ZK_FR_WORDS 4
#define ZK_G1_WORDS 8
#define ZK_G2_WORDS 16

/* ------------------------------------------------------------------------------------------
 * Context
 * ---------------------------------------------------------------------------------------- */
int zk_ctx_create(int device_ordinal, zk_ctx** out);
void zk_ctx_destroy(zk_ctx* ctx);
const char* zk_strerror(int status);
const char* zk_last_error(const zk_ctx* ctx);          /* detail of the last failing call */
/* Options of the product build: "msm_window_bits" (Pippenger c of the fixed-base tables; 0 = automatic, c = the same for every table,
 * 100 * big + small = `big` for tables of 2^21 points and more, + 10000 * g2 = its own window for the G2 table), "msm_shard_points"
 * (zk_prove_partial: 0 = a rank owns Pippenger windows, 1 = a rank owns a range of the points, 2 = a rank owns 1 / world of the BUCKET
 * range of every product -- the fixed-base tables share one bucket set over all windows, so this divides entries, accumulation and
 * the per-bucket reduction tail by world where 13 windows over 8 ranks leave 2 : 1; world a power of two, otherwise as 0), "rank_tables" (multi-GPU exchange:
 * 1 = window tables of the rank's own point ranges only), "dense_long_division" (1: the dense form always divides by t with the
 * reference's long division; default 0 = power-series inverse above 512 quotient coefficients), "msm_quad_buckets" (inner products of
 * at most this many buckets run their reduction tail with four lanes per point addition: shorter dependency chains for small
 * circuits; default 65536, 0 = never), "interp_large_log" (arbitrary-roots interpolation: trees of at least 2^value coefficients per
 * level take the form that re-uses the children's transforms; default 20; both forms give the same coefficients), "comm_cu_reserve"
 * (zk_mgpu_create over an RCCL communicator of more than one rank: compute units per XCD that the inner-product streams leave to the
 * collectives' kernels; default 0 = none: measured on one GPU it costs 4 % of the rate and shortens the p90 wait of an all-to-all only
 * from 2.7 to 2.2 ms, profiles/r4_rccl_starvation.txt), "basis_tree_min" (see zk_crs_upload), "merge_lh" (default 1: the witness
 * product L = sum a_i sum_delta_i and H + r B1 + s A, which only occur added together in the proof element c, are ONE inner product
 * over the table xi_t | xi | sum_delta with one set of buckets and one reduction tail; 0 = two products as in round 4; same proof
 * bytes, +1.3 % proofs/s at 2^20 gates, profiles/r5_experiments.txt item 2).  Each is exercised by a -m gpu test.  Unknown keys are
 * answered with ZK_ERR_UNSUPPORTED.  Measurement entry points and switches -- kernel event timing, the tuning keys of bench.py --opt /
 * --serialize -- are NOT part of this header: include/zkgpu_measure.h. */
int zk_set_option(zk_ctx* ctx, const char* key, long value);
long zk_get_option(const zk_ctx* ctx, const char* key);

/* ------------------------------------------------------------------------------------------
 * Building blocks (individually parity-tested against the oracle)
 * ---------------------------------------------------------------------------------------- */
/* Radix-2 NTT over Fr, natural order in and out, in place on a host buffer of 2^log_n elements.
 *   inverse=0: out[k] = sum_j in[j] * w^(jk)            == field::dft   (field/mod.rs:508-520)
 *   inverse=1: w^-1 and scaling by n^-1                 == field::idft  (field/mod.rs:524-537)
 * with w = 5^((r-1)/2^log_n).  coset=1 evaluates on / interpolates from the coset g*<w> with
 * g = 5^((r-1)/2^(log_n+1)) (forward: in[j] *= g^j first; inverse: out[j] *= g^-j last).  log_n <= 24 (23 with coset). */
int zk_ntt_fr(zk_ctx* ctx, uint64_t* data, unsigned log_n, int inverse, int coset);
/* Interpolation through ARBITRARY distinct nodes: coeffs[0..n) = the coefficients of the polynomial of degree < n with
 * p(roots[k]) = values[k] -- what QAP::from does per wire polynomial with Lagrange sums (fr.rs:140-173, coefficient_poly.rs:159-200;
 * O(n^2) each) and the prover of an arbitrary-roots QAP (zk_qap_upload_sparse_roots) does per proof for U and V,
 * by a sub-product tree over batched NTTs in O(n log^2 n) (csrc/interp.hip).  1 <= n <= 2^23; ZK_ERR_ARG when two roots
 * coincide. */
int zk_interpolate_fr(zk_ctx* ctx, const uint64_t* roots, const uint64_t* values, size_t n, uint64_t* coeffs);

/* sum_i scalars[i] * points[i]: the SigmaG1/SigmaG2 inner products of groth16::prove
 * (groth16/mod.rs:255-272,279-290), i.e. n x exp_encrypted_g1/g2 (fr.rs:114-119) folded with
 * Sum for G1Local/G2Local (fr.rs:191-198,217-223).  window_bits = 0 picks automatically.  zk_msm_g1 only: window_bits in
 * [-9, -2] runs the Pippenger form BASELINE.json's north_star words with c = -window_bits -- one wavefront per (window, chunk),
 * buckets in LDS, a wave-level fold of the partial sums (csrc/msm_lds.hpp) -- a measured comparator, never used by zk_prove. */
/* Window size c (bits) the fixed-base tables of a product of `count` points are built with when the option msm_window_bits
 * is 0: the outcome of the sweeps in DESIGN.md 4c (G1: 17 from 2^17 points, 20 from 2^21; the G2 table: 20 from 2^20).  Host code, no
 * device needed.  (The reference has no such notion: fr.rs:114-119 is one double-and-add per term.) */
int zk_msm_auto_window(size_t count);
int zk_msm_auto_window_g2(size_t count);
int zk_msm_g1(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, int window_bits, uint64_t out_affine[ZK_G1_WORDS]);
int zk_msm_g2(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, int window_bits, uint64_t out_affine[ZK_G2_WORDS]);

/* Element-wise field / group kernels (diagnostic entry points used by the parity tests).
 * op: 0 add, 1 sub, 2 mul, 3 inverse of a (b ignored; ZK_ERR_DIV_BY_ZERO if any a == 0), 4 the same inverse by the
 * binary extended Euclid (ff.cuh inv_euclid), 5 by division steps in batches of 30 (inv_divsteps: what closes a proof)
 * FrLocal Add/Sub/Mul/Div: fr.rs:18-71 */
int zk_fr_batch(zk_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zk_fq_batch(zk_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* out[i] = scalars[i] * points[i]  (exp_encrypted_g1 / exp_encrypted_g2, fr.rs:114-119) */
int zk_g1_mul_batch(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t* out, size_t n);
int zk_g2_mul_batch(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t* out, size_t n);
/* out[i] = a[i] + b[i]  (Add for G1Local/G2Local, fr.rs:175-215) */
int zk_g1_add_batch(zk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zk_g2_add_batch(zk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

/* ------------------------------------------------------------------------------------------
 * QAP  (QAP<CoefficientPoly<FrLocal>>, groth16/mod.rs:60-67; built by fr.rs:140-173)
 * ---------------------------------------------------------------------------------------- */
/* One of u, v, w in root representation (circuit::RootRepresentation, circuit/mod.rs:201-214;
 * DummyRep rows, dummy_rep.rs:6-13): CSR by WIRE, entry k of wire i = (gate index, value),
 * meaning polynomial_i(root[gate]) = value.  Duplicate (wire, gate) entries add up, as the
 * reference's Lagrange sum does (coefficient_poly.rs:159-171). */
typedef struct {
    const uint64_t* ptr;   /* m+1 offsets */
    const uint32_t* gate;  /* nnz gate indices in [0, n) */
    const uint64_t* val;   /* nnz Fr values (4 words each) */
} zk_sparse_rows;

/* Sparse QAP on the domain roots[j] = w^j, w = 5^((r-1)/n), n = 2^log_n gates: the large-circuit
 * form.  Equivalent to QAP::from(root_rep) (fr.rs:140-173) with those roots; the dense
 * per-wire polynomials are never materialised. t(x) = x^n - 1. */
typedef struct {
    unsigned log_n;
    size_t m;       /* wires; wire 0 is the constant 1 */
    size_t input;   /* l = number of verifier-supplied wires (qap.input) */
    zk_sparse_rows u, v, w;
} zk_qap_sparse_desc;
int zk_qap_upload_sparse(zk_ctx* ctx, const zk_qap_sparse_desc* desc, zk_qap** out);
/* The same rows over the roots ASTParser emits, the integers 1..n (circuit/mod.rs:517: `roots: (1..=n)`), for any
 * n <= 2^23 (desc->log_n is ignored).  Equivalent to QAP::from(root_rep) (fr.rs:140-173) followed by the reference's
 * coefficient-form prove (mod.rs:199-290) -- the proofs are byte-identical to the dense form's -- but nothing is ever
 * interpolated: the prover keeps U, V as values on {1..n}, the quotient as values on {n+1..2n-1}, and takes its inner
 * products with the CRS in those Lagrange bases, which zk_setup emits next to the reference's [x^i] arrays.  Costs
 * O(nnz + n log n) per proof where the dense form is O(m n); SURVEY.md 8-f4.  With a CRS that carries only the reference's arrays
 * (zk_crs_upload, a ZKCRSv1 file) the first proof derives the Lagrange-basis points from [x^i], once per CRS: from 16384 gates on
 * (option "basis_tree_min") by the transpose of the interpolation tree run over curve points, O(n log^2 n) point operations (2.2 s at
 * 2^16 gates, 40 s at 2^20, n <= 2^22; csrc/gbasis.hip), below by n^2 inner products (csrc/basis.hip).  With basis_tree_min < 0 a
 * CRS of more than 2^16 + 2^10 gates is served through the form of zk_qap_upload_sparse_roots with the roots 1..n instead (same
 * bytes, 0.6 x the rate).
 * Batches and the multi-GPU entry points take both sparse forms. */
int zk_qap_upload_sparse_integers(zk_ctx* ctx, const zk_qap_sparse_desc* desc, size_t n, zk_qap** out);
/* The same rows over ANY distinct roots r_0 .. r_{n-1} the caller supplies (RootRepresentation::roots() is caller data,
 * circuit/mod.rs:201-214, dummy_rep.rs:47): gate j = root roots[j] (4 words each, canonical), 1 <= n <= 2^22.  Equivalent to
 * QAP::from(root_rep) (fr.rs:140-173) followed by the reference's coefficient-form prove -- byte-identical proofs -- but the 3 m wire
 * polynomials are never interpolated: the prover interpolates U = sum a_i u_i and V per proof from their values on the roots by a
 * sub-product tree (csrc/interp.hip, O(n log^2 n)), divides U V by t (the remainder is dropped) and takes its inner products with the
 * reference's own [x^i] arrays, so ANY CRS for the circuit serves (zk_setup, zk_crs_upload, a file).  The tables of a root set (the tree,
 * the weights 1 / N'(r_k) by a scaled remainder tree) are O(n log^2 n) too: 0.14 s at 2^20 gates.  ZK_ERR_ARG when two roots coincide.  One proof at a time or pipelined on one GPU, or
 * window-sharded (zk_prove_partial); batches and the scalar exchange take the two forms above (ZK_ERR_UNSUPPORTED). */
int zk_qap_upload_sparse_roots(zk_ctx* ctx, const zk_qap_sparse_desc* desc, const uint64_t* roots, size_t n, zk_qap** out);

/* Dense coefficient form, exactly the fields of QAP<CoefficientPoly<FrLocal>>: u, v, w are
 * m x n row-major (coefficient k of wire i at [i*n + k], zero padded), t has n+1 coefficients
 * (any roots; e.g. ASTParser's 1..n, circuit/mod.rs:517). */
int zk_qap_upload_dense(zk_ctx* ctx, const uint64_t* u, const uint64_t* v, const uint64_t* w, const uint64_t* t,
                        size_t m, size_t n, size_t input, zk_qap** out);
void zk_qap_free(zk_qap* qap);

/* Dimensions of a device QAP, and its dense coefficient matrices back on the host (dense form only;
 * any pointer may be NULL = skip).  Layout as zk_qap_upload_dense. */
int zk_qap_dims(const zk_qap* qap, size_t* n, size_t* m, size_t* input, int* dense);
/* Which device form a QAP handle holds (a container read by zk_qap_load does not say otherwise): 0 = sparse rows over the roots of
 * unity w^j, n = 2^k (zk_qap_upload_sparse); 1 = the dense m x n coefficient matrices of QAP<CoefficientPoly<FrLocal>>
 * (groth16/mod.rs:60-67; zk_qap_upload_dense, zk_circuit_qap); 2 = sparse rows over the integers 1..n, the roots ASTParser emits
 * (circuit/mod.rs:517; zk_qap_upload_sparse_integers); 3 = sparse rows over the caller's roots (zk_qap_upload_sparse_roots). */
int zk_qap_kind(const zk_qap* qap);
/* The first stage of groth16::prove on its own (diagnostic entry point of the parity tests): u_sum = sum_i qap.u[i] * weights[i]
 * (groth16/mod.rs:233-253 -- zip with the weights, CoefficientPoly: Mul<T> coefficient_poly.rs:132-146, Sum :75-91); which = 0 / 1 / 2
 * for u / v / w.  Dense form: out = the n coefficients of the sum.  Sparse forms: out = its n values on the QAP's domain (w^j or
 * j + 1, j < n), u and v only (the prover never evaluates W).  4 words per element, canonical; weights beyond m_qap are ignored. */
int zk_qap_weighted_sum(zk_ctx* ctx, const zk_qap* qap, const uint64_t* weights, size_t m, int which, uint64_t* out);
int zk_qap_download_dense(zk_ctx* ctx, const zk_qap* qap, uint64_t* u, uint64_t* v, uint64_t* w, uint64_t* t);

/* ------------------------------------------------------------------------------------------
 * .zk front end (the input side of the path; host code, as in the reference)
 * ---------------------------------------------------------------------------------------- */
typedef struct zk_circuit zk_circuit;   /* DummyRep<FrLocal> (circuit/dummy_rep.rs:6-13) + the parsed program */
/* ASTParser::try_parse (circuit/mod.rs:224-527; tokenizer/AST: circuit/ast.rs).  On a parse error
 * returns ZK_ERR_ARG and writes the ParseErr text ("SyntaxErr(line, ..)" / "StructureErr(gate, ..)")
 * to err.  Wire 0 is the constant 1, then the `verify` variables, then first appearance; gate k
 * (1-based) sits at root k (circuit/mod.rs:517). */
int zk_circuit_parse(const char* code, zk_circuit** out, char* err, size_t err_len);
void zk_circuit_free(zk_circuit* c);
int zk_circuit_dims(const zk_circuit* c, size_t* m, size_t* n, size_t* input, size_t* n_in);
/* Root representation rows (RootRepresentation::u/v/w, circuit/mod.rs:201-214): which = 0 u, 1 v, 2 w;
 * ptr[m+1], gate[nnz] (0-based gate index = root - 1), val[nnz*4].  NULL arrays are skipped. */
int zk_circuit_rows(const zk_circuit* c, int which, uint64_t* ptr, uint32_t* gate, uint64_t* val, size_t* nnz);
/* circuit::weights (circuit/mod.rs:529-637): inputs in `in` order -> [1] ++ wire values (m x 4 words) */
int zk_circuit_weights(const zk_circuit* c, const uint64_t* inputs, size_t n_in, uint64_t* weights_out, size_t m);
const char* zk_circuit_last_error(const zk_circuit* c);
/* QAP<CoefficientPoly<FrLocal>>::from(root_rep) (fr.rs:140-173; Lagrange interpolation
 * coefficient_poly.rs:159-200) on the GPU for the circuit's roots 1..n -> dense device QAP. */
int zk_circuit_qap(zk_ctx* ctx, const zk_circuit* c, zk_qap** out);
/* The same QAP through zk_qap_upload_sparse_integers: no interpolation, no 16384-gate limit, identical proofs. */
int zk_circuit_qap_sparse(zk_ctx* ctx, const zk_circuit* c, zk_qap** out);

/* ------------------------------------------------------------------------------------------
 * CRS  (SigmaG1 / SigmaG2, groth16/mod.rs:105-121)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t n, m, input;
    const uint64_t *alpha_g1, *beta_g1, *delta_g1;   /* 8 words each */
    const uint64_t* xi_g1;          /* n       points  [x^i]_1            */
    const uint64_t* sum_gamma_g1;   /* input+1 points  (wires 0..l)       */
    const uint64_t* sum_delta_g1;   /* m-l-1   points  (wires l+1..m-1)   */
    const uint64_t* xi_t_g1;        /* n-1     points  [x^i t(x)/delta]_1 */
    const uint64_t *beta_g2, *gamma_g2, *delta_g2;   /* 16 words each */
    const uint64_t* xi_g2;          /* n       points  [x^i]_2            */
} zk_crs_desc;
int zk_crs_upload(zk_ctx* ctx, const zk_crs_desc* desc, zk_crs** out);

/* groth16::setup (groth16/mod.rs:134-197) on the GPU with the five random draws injected:
 * trapdoor = alpha | beta | gamma | delta | x (4 words each, all non-zero). */
int zk_setup(zk_ctx* ctx, const zk_qap* qap, const uint64_t trapdoor[20], zk_crs** out);

/* Copy a device CRS back into caller-provided host buffers (any pointer may be NULL = skip). */
typedef struct {
    uint64_t *alpha_g1, *beta_g1, *delta_g1, *xi_g1, *sum_gamma_g1, *sum_delta_g1, *xi_t_g1;
    uint64_t *beta_g2, *gamma_g2, *delta_g2, *xi_g2;
} zk_crs_out;
int zk_crs_dims(const zk_crs* crs, size_t* n, size_t* m, size_t* input);
int zk_crs_download(zk_ctx* ctx, const zk_crs* crs, const zk_crs_out* out);
void zk_crs_free(zk_crs* crs);

/* On-disk CRS container.  The reference has no serialisation of SigmaG1/SigmaG2
 * (groth16/mod.rs:105-121), and setup (mod.rs:134-197) draws a fresh trapdoor on every call, so a CRS must be
 * written down to be reused.  Format: "ZKCRSv1\0", n, m, input, FNV-1a-64 of the payload, then the arrays of
 * zk_crs_desc in declaration order as canonical little-endian words.  A CRS that zk_setup made for an integer-roots QAP
 * (zk_qap_upload_sparse_integers) is written as "ZKCRSv2\0": the same, followed by its Lagrange-basis arrays, so that the
 * reloaded CRS serves that QAP form again.  zk_crs_load reads both, range- and curve-checks every point
 * and returns ZK_ERR_IO for a missing, truncated or altered file. */
int zk_crs_save(zk_ctx* ctx, const zk_crs* crs, const char* path);
int zk_crs_load(zk_ctx* ctx, const char* path, zk_crs** out);
/* The same for a QAP ("ZKQAPv1": the sparse rows over the roots w^j, or the dense coefficient matrices -- what
 * zk_qap_upload_sparse / zk_qap_upload_dense take; QAP<P> has no serialisation in the reference, groth16/mod.rs:60-67) and for a
 * proof ("ZKPRFv1": versioned magic | the 259 canonical bytes | checksum; Proof has no encoding in the reference, mod.rs:124-128).
 * ZK_ERR_IO for missing, truncated, altered or foreign files; values are range-checked by the upload path. */
int zk_qap_save(zk_ctx* ctx, const zk_qap* qap, const char* path);
int zk_qap_load(zk_ctx* ctx, const char* path, zk_qap** out);
int zk_proof_save(const uint8_t proof[ZK_PROOF_BYTES], const char* path);
int zk_proof_load(const char* path, uint8_t proof_out[ZK_PROOF_BYTES]);

/* ------------------------------------------------------------------------------------------
 * prove  (groth16::prove, groth16/mod.rs:213-296) with (r, s) injected (mod.rs:231)
 * ---------------------------------------------------------------------------------------- */
int zk_prove(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const uint64_t* weights, size_t m,
             const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]);
/* Same with the witness already resident in HBM (m x 4 words, canonical form). */
int zk_prove_dev(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                 const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]);

/* Pipelined form of zk_prove_dev for a stream of proofs: zk_prove_submit enqueues the whole proof
 * and returns at once with a ticket; zk_prove_wait blocks until that proof's bytes are ready.  At most
 * ZK_MAX_IN_FLIGHT proofs may be in flight per context (a further submit returns ZK_ERR_ARG until one is waited
 * for); the next proof's witness products and NTT stage then run under the previous one's reduction tail.  Two
 * in flight saturate one GPU on whole proofs; the short per-rank shares of a sharded proof profit from four.
 * The witness buffer must stay valid and unmodified until the matching wait.  Errors that are only
 * detected on the device (witness element >= r) are reported by zk_prove_wait. */
int zk_prove_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                    const uint64_t r[4], const uint64_t s[4], int* ticket);
int zk_prove_wait(zk_ctx* ctx, int ticket, uint8_t* proof_out /* ZK_PROOF_BYTES; may be NULL for a partial ticket */);
/* zk_prove_submit with the witness in HOST memory (what a caller of groth16::prove holds, mod.rs:213-217): the 32 m
 * bytes are copied into a device buffer owned by the ticket, on the stream the proof starts on.  With page-locked
 * memory from zk_host_alloc the copy is asynchronous and the transfer of one proof overlaps the inner products of the
 * previous one (PCIe-inclusive rate of a stream of proofs = resident rate); with pageable memory the call blocks for
 * the copy.  The buffer must stay valid and unmodified until the matching zk_prove_wait. */
int zk_prove_submit_host(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const uint64_t* weights, size_t m,
                         const uint64_t r[4], const uint64_t s[4], int* ticket);
int zk_host_alloc(size_t bytes, void** out);   /* page-locked host memory (hipHostMalloc); no counterpart in the reference, whose
                                                * weights live in a Vec<FrLocal> (lib.rs:107) */
void zk_host_free(void* p);

/* Batches (roots-of-unity QAP form): what `count` calls of groth16::prove (groth16/mod.rs:213-296) over the same QAP and
 * CRS compute -- `count` proofs with their own witnesses and (r, s) -- as one unit of work -- the SpMV / NTT stages follow each other, the inner products of all proofs run as one grouped MSM per
 * product, two launches assemble them.  For circuits of 2^16 gates and fewer a single proof is bound by the latency
 * of its ~100 dependent launches; a batch spreads that chain over `count` proofs (2^16 gates: 2.6 ms per proof alone).
 * d_weights[j]: device pointer to proof j's witness (m[j] x 4 words, canonical); r, s: count x 4 words;
 * proofs_out: count x ZK_PROOF_BYTES.  A batch ticket counts as one proof in flight; a witness element >= r fails
 * the whole batch (ZK_ERR_RANGE from zk_prove_batch_wait). */
#define ZK_MAX_BATCH 64
int zk_prove_batch_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, int count, const void* const* d_weights, const size_t* m,
                          const uint64_t* r, const uint64_t* s, int* ticket);
int zk_prove_batch_wait(zk_ctx* ctx, int ticket, int count, uint8_t* proofs_out);

/* Multi-GPU: every rank holds the CRS and recomputes the NTT stage; rank g owns
 * Pippenger windows w = g (mod world) of each inner product and writes its partial sums
 * (Jacobian, device Montgomery limbs) to d_partial_out (ZK_PARTIAL_BYTES).  The caller all-gathers
 * the blobs (RCCL, as bytes) and any rank finishes with zk_prove_combine.  (r, s) are needed here
 * because B in G1 enters the proof only as r*B1 and is folded into the H product as scalars r*v_i. */
#define ZK_PARTIAL_BYTES 768   /* 4 G1 Jacobian slots (96 B) + 1 G2 Jacobian (192 B), padded to 768 */
int zk_prove_partial(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                     const uint64_t r[4], const uint64_t s[4], int rank, int world, void* d_partial_out);
int zk_prove_combine(zk_ctx* ctx, const zk_crs* crs, const void* d_partials, int world,
                     const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]);
/* Pipelined form of zk_prove_partial (same in-flight rule as zk_prove_submit); finish with
 * zk_prove_wait(ctx, ticket, NULL), after which d_partial_out holds the rank's partial sums. */
int zk_prove_partial_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                            const uint64_t r[4], const uint64_t s[4], int rank, int world, void* d_partial_out, int* ticket);

/* Multi-GPU, scalar exchange (roots-of-unity QAP form only).  Instead of repeating the SpMV / NTT
 * stage on every rank, the ranks take turns: in a round of `world` proofs rank j runs that stage for proof j
 * (zk_prove_scalars_submit) and writes the scalars of the four inner products (groth16/mod.rs:255-290: L over
 * sum_delta, V over [x^i]_2, U over [x^i]_1, h | r v + s u over [x^i t/delta]_1 | [x^i]_1) as `world` equal
 * chunks each -- chunk g multiplies the points rank g owns, [g c, (g+1) c) with c = ceil(count / world), zero
 * scalars behind the last point.  One all-to-all per array (RCCL, equal splits) hands every rank its chunk of
 * every proof of the round; zk_prove_msm_submit accumulates them over the rank's points (partial sums of proof j
 * at d_partials_out + j * ZK_PARTIAL_BYTES), a second all-to-all returns the blobs to the proofs' owners, and
 * zk_prove_combine finishes.  Each array has elems_out[k] elements of 32 bytes (k = L, V, U, H). */
int zk_prove_exchange_elems(const zk_qap* qap, int world, size_t elems_out[4]);
int zk_prove_scalars_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                            const uint64_t r[4], const uint64_t s[4], int world,
                            void* d_l, void* d_v, void* d_u, void* d_h, int* ticket);
/* the same from a host witness (see zk_prove_submit_host) */
int zk_prove_scalars_submit_host(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const uint64_t* weights, size_t m,
                                 const uint64_t r[4], const uint64_t s[4], int world,
                                 void* d_l, void* d_v, void* d_u, void* d_h, int* ticket);
/* d_l .. d_h: `sets` chunks each, as delivered by the all-to-all (chunk j = proof j's scalars for this rank's points).
 * One ticket for the batch; finish with zk_prove_wait(ctx, ticket, NULL). */
int zk_prove_msm_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, int sets, int rank, int world,
                        const void* d_l, const void* d_v, const void* d_u, const void* d_h, void* d_partials_out, int* ticket);

/* ------------------------------------------------------------------------------------------
 * Several GPUs ("one context owns 1..8 devices"; 8e).  The reference's groth16::prove
 * (groth16/mod.rs:213-217) is one call on one thread; its inner products (mod.rs:255-272,279-290) are sums of
 * independent terms and shard over GPUs.  One process per GPU, each with its own zk_ctx; a zk_comm joins them: RCCL over
 * xGMI (libzkgpu.so links librccl), or a transport supplied by the caller as function pointers.
 * ---------------------------------------------------------------------------------------- */
int zk_device_count(void);
#define ZK_COMM_ID_BYTES 128
typedef struct zk_comm zk_comm;
/* Rank 0 draws the id (ncclGetUniqueId) and ships the bytes to the other ranks by any channel (file, TCP store, MPI). */
int zk_comm_unique_id(uint8_t id_out[ZK_COMM_ID_BYTES]);
/* Collective: every rank calls it with the same id and its own rank (ncclCommInitRank on ctx's device).  world == 1
 * needs no id and no RCCL.  The set-up is bounded like every other wait on a peer (ZK_COMM_TIMEOUT_MS): ncclCommInitRank runs on a
 * helper thread, and when a peer never arrives the call returns ZK_ERR_COMM while that thread stays inside RCCL's bootstrap.  After
 * such a time-out do NOT retry with the same id in this process and do not unload the library: report the failure and let the
 * process end (bench.py's `degraded` path does exactly that). */
int zk_comm_init(zk_ctx* ctx, const uint8_t id[ZK_COMM_ID_BYTES], int rank, int world, zk_comm** out);
/* Caller-supplied transport.  Buffers are the ones the prover allocated (device memory with the GPU backend); the calls
 * are blocking: complete on return.  all_to_all: chunk g of `send` (bytes_per_rank bytes) goes to rank g, chunk j of `recv`
 * comes from rank j.  all_gather: `send` (bytes_per_rank) from every rank, in rank order, into `recv`.  barrier / max_f64
 * may be NULL (zk_comm_barrier / zk_comm_max_f64 then return ZK_ERR_UNSUPPORTED).  Non-zero return = failure. */
typedef struct {
    void* user;
    int (*all_to_all)(void* user, const void* send, void* recv, size_t bytes_per_rank);
    int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank);
    int (*barrier)(void* user);
    int (*max_f64)(void* user, double* value);
} zk_comm_ops;
int zk_comm_init_custom(zk_ctx* ctx /* may be NULL */, const zk_comm_ops* ops, int rank, int world, zk_comm** out);
/* A communicator outlives the zk_mgpu provers created over it.  Destroying it while a prover still holds it is safe all the same:
 * the call then only marks it, and the last zk_mgpu_destroy frees it. */
void zk_comm_destroy(zk_comm* comm);
int zk_comm_rank(const zk_comm* comm);
int zk_comm_world(const zk_comm* comm);
/* ranks of the RCCL communicator behind this zk_comm as RCCL counts them (ncclCommCount); 0 = none (one rank, a caller's transport,
 * aborted).  Evidence for the bench line that a world-N run really went through an N-rank RCCL communicator. */
int zk_comm_rccl_ranks(const zk_comm* comm);
/* Every host wait for a collective (zk_comm_barrier / _max_f64 / _all_to_all / _all_gather, zk_mgpu_pop) is bounded: after `ms`
 * milliseconds (default 120000, or ZK_COMM_TIMEOUT_MS at zk_comm_init; 0 = unbounded) the RCCL communicator is aborted and the call
 * returns ZK_ERR_COMM, as does every later one.  The reference's prove (mod.rs:213-217) cannot hang on a peer; neither may this. */
int zk_comm_set_timeout(zk_comm* comm, long ms);
/* Gives up on the peers now (ncclCommAbort): callable from another thread while a collective is being waited for OR enqueued -- an
 * enqueue that is under way finishes its RCCL calls first (the abort waits for it, 200 ms at most); every later call answers
 * ZK_ERR_COMM. */
int zk_comm_abort(zk_comm* comm);
int zk_comm_barrier(zk_comm* comm);
int zk_comm_max_f64(zk_comm* comm, double* value);      /* *value = max over the ranks (timing of the slowest rank) */
int zk_comm_all_to_all(zk_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank);   /* complete on return */
int zk_comm_all_gather(zk_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank);

/* Latency form -- ONE groth16::prove over all ranks: every rank holds the same witness, recomputes the SpMV / NTT stage,
 * accumulates its share of the inner products (three with merge_lh; zk_prove_partial: Pippenger windows w = rank (mod world), or point
 * ranges with the option msm_shard_points), one 768-byte all-gather, and every rank assembles the same 259 bytes. */
int zk_mgpu_prove_sharded(zk_ctx* ctx, zk_comm* comm, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                          const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]);

/* Throughput form -- the scalar exchange of zk_prove_scalars_submit / zk_prove_msm_submit as a software pipeline inside the
 * library.  A ROUND is `world` proofs, one per rank: zk_mgpu_push hands in THIS rank's proof of the next round (its own
 * witness and (r, s)), zk_mgpu_pop returns THIS rank's proof of the oldest round.  Both are collective: every rank makes
 * the same sequence of push / pop calls.  At most three rounds may be pushed and not yet popped; pushing two rounds ahead
 * of every pop (push, push, push, pop, push, pop, ...) keeps the SpMV / NTT stage two rounds ahead of the inner products, so
 * that neither the exchanges nor the host waits leave a GPU idle; push / pop strictly alternating is the one-round-at-a-time
 * (latency) schedule.  Per-GPU work per round is one whole proof's worth whatever `world` is.  Sparse QAP forms only
 * (roots of unity or integer roots). */
typedef struct zk_mgpu zk_mgpu;
int zk_mgpu_create(zk_ctx* ctx, zk_comm* comm, const zk_crs* crs, const zk_qap* qap, zk_mgpu** out);
int zk_mgpu_push(zk_mgpu* p, const void* d_weights, size_t m, const uint64_t r[4], const uint64_t s[4]);
/* The same with the witness in host memory, as the reference's prove(&[T]) hands it over (mod.rs:213-217): copied into a
 * device buffer owned by the round on the stream the proof starts on (page-locked memory from zk_host_alloc: asynchronously). */
int zk_mgpu_push_host(zk_mgpu* p, const uint64_t* weights, size_t m, const uint64_t r[4], const uint64_t s[4]);
int zk_mgpu_pop(zk_mgpu* p, uint8_t proof_out[ZK_PROOF_BYTES]);
void zk_mgpu_destroy(zk_mgpu* p);
const char* zk_mgpu_last_error(const zk_mgpu* p);
/* The pipeline over caller-supplied stages (tests: CPU stand-ins + a gloo transport exercise the schedule and the order of
 * the collectives without a GPU).  elems: sizes of the four exchange arrays in 32-byte elements (multiples of world);
 * scalars_submit writes `world` equal chunks into send[0..3]; msm_submit consumes recv[0..3] (chunk j = proof j) and writes
 * `sets` blobs of ZK_PARTIAL_BYTES; wait completes a ticket; combine assembles from the `world` blobs returned to the owner. */
typedef struct {
    void* user;
    int (*elems)(void* user, int world, size_t elems_out[4]);
    void* (*alloc)(void* user, size_t bytes);
    void (*free)(void* user, void* p);
    int (*scalars_submit)(void* user, const void* weights, size_t m, const uint64_t r[4], const uint64_t s[4], int world, void* const send[4], int* ticket);
    int (*msm_submit)(void* user, int sets, int rank, int world, void* const recv[4], void* partials_out, int* ticket);
    int (*wait)(void* user, int ticket);
    int (*combine)(void* user, const void* partials, int world, const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]);
} zk_mgpu_backend;
int zk_mgpu_create_custom(zk_comm* comm, const zk_mgpu_backend* backend, zk_mgpu** out);

/* ------------------------------------------------------------------------------------------
 * verify  (groth16::verify, groth16/mod.rs:299-320) -- host code, as in the reference
 * ---------------------------------------------------------------------------------------- */
/* *ok = 1 iff e(alpha,beta) e(sum_i x_i sum_gamma_i, gamma) e(C,delta) == e(A,B) with x = (1, inputs...).
 * inputs: n_inputs Fr values (the `verify` wires, without the leading 1).  A malformed proof gives *ok = 0: the
 * decoder accepts exactly one byte string per point (tag 0x00 followed by zeros only = infinity; tag 0x04 + coordinates
 * < q on the curve, never (0, 0)) and B must lie in the order-r subgroup G2 of the twist ([r]B = infinity is checked:
 * the twist's cofactor has small factors and the pairing is bilinear only on G2). */
int zk_verify(zk_ctx* ctx, const zk_crs* crs, const uint64_t* inputs, size_t n_inputs, const uint8_t proof[ZK_PROOF_BYTES], int* ok);
/* EllipticEncryptable::pairing (fr.rs:120-122): the optimal ate pairing e(P, Q) as 12 Fq coefficients
 * (48 words) in the order c0.a0.c0, c0.a0.c1, c0.a1.c0, ..., c1.a2.c1 of the tower
 * Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - (9+i)).  Host only; needs no context.  ZK_ERR_RANGE when a coordinate
 * is >= q, a point is off its curve or g2 is outside the order-r subgroup. */
int zk_pairing(const uint64_t g1[ZK_G1_WORDS], const uint64_t g2[ZK_G2_WORDS], uint64_t out[48]);

#ifdef __cplusplus
}
#endif
#endif /* ZKGPU_H */
