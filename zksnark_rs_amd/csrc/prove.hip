// prove.hip -- groth16::prove (/root/reference/src/groth16/mod.rs:213-296) as a GPU pipeline.
//
//   reference step (mod.rs)                         here
//   ---------------------------------------------   -----------------------------------------
//   u_sum/v_sum/w_sum = sum qap.*[i]*a_i  :233-253   SpMV over gates -> evaluations; inverse NTT
//   a_g1,b_g1,b_g2 = sum coeff_i * xi_i   :255-272   Pippenger MSM (G1, G1, G2)
//   h = (u_sum*v_sum - w_sum)/t           :277       t = x^n-1: quotient == high half of U*V;
//                                                     obtained from evaluations on <w> and on the
//                                                     coset g<w> (g^n = -1):
//                                                       lo+hi = iNTT(U.V on <w>), lo-hi = coset-iNTT(U.V on g<w>)
//                                                     => identical to the reference's quotient for
//                                                     EVERY witness (also unsatisfying ones).
//   sum h_i * xi_t_i, sum a_i * sum_delta :279-290   Pippenger MSM (G1, G1)
//   a, b, c assembly with r, s            :274-293   one small kernel; scalar mults by r, s
//
// (r, s) are injected: the reference draws them from thread_rng (mod.rs:231).
#include <algorithm>
#include <cstring>
#include <functional>
#include <random>
#include <type_traits>
#include <vector>
#include "pipeline.hpp"
#include "qap_kernels.hpp"
#include "lazy29.cuh"
#include <sys/random.h>

namespace zk {

void crs_ensure_brev(zk_ctx* ctx, zk_crs& c, unsigned log_n);

struct MsmResults {
    G1J a;      // sum u_i [x^i]_1
    G1J hb;     // sum h_i [x^i t/delta]_1 + sum (r v_i + s u_i) [x^i]_1   (H, r*B1 and s*A only occur in c)
    G1J l;      // sum a_i sum_delta_i
    G1J spare;
    G2J b2;
};
static_assert(sizeof(MsmResults) == 4 * 96 + 192, "partial layout");
static_assert(sizeof(MsmResults) <= ZK_PARTIAL_BYTES, "ZK_PARTIAL_BYTES too small");

struct AssemblePre {
    G1J r_delta;      // alpha1 + r * delta1
    G1J fixed_c;      // s * alpha1 + r * beta1 + (r s) * delta1
    G2J s_delta2;     // beta2 + s * delta2
};

__device__ __forceinline__ uint32_t nibble(const Fr& k, int w) { return (k.l[w >> 3] >> ((w & 7) * 4)) & 15u; }

// k * P from the 4-bit fixed-base table FT[w][d] = d * 16^w * P: lane w contributes FT[w][digit_w]; the 64
// contributions of a wave are summed by a tree over the wave's own LDS row (6 additions deep instead of 254 doublings).
// Everything that depends only on (r, s) and single CRS points; runs on the side stream while the five inner products
// execute.  One wave per fixed-base multiplication: waves 0..3 in G1 (r delta, s alpha, r beta, rs delta), wave 4 in G2
// (s delta2).  All waves run the SAME loop with workgroup barriers at uniform places -- the G1 / G2 difference is inside
// barrier-free regions -- so no barrier sits in divergent control flow.
__device__ __forceinline__ void assemble_pre_body(const G1A* __restrict__ ft_alpha1, const G1A* __restrict__ ft_beta1, const G1A* __restrict__ ft_delta1,
                                                  const G2A* __restrict__ ft_delta2, const G1A* __restrict__ alpha1, const G2A* __restrict__ beta2,
                                                  const Fr& r, const Fr& s, AssemblePre* __restrict__ out) {
    __shared__ G1J sh1[4][64];
    __shared__ G2J sh2[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const Fr rs = (Fr::from_canonical(r) * Fr::from_canonical(s)).to_canonical();
    const Fr k = wave == 0 ? r : wave == 1 ? s : wave == 2 ? r : wave == 3 ? rs : s;
    const G1A* ft1 = wave == 0 ? ft_delta1 : wave == 1 ? ft_alpha1 : wave == 2 ? ft_beta1 : ft_delta1;
    const uint32_t digit = nibble(k, lane);
    if (wave < 4) sh1[wave][lane] = G1J::from_affine(ft1[lane * 16 + digit]);
    else sh2[lane] = G2J::from_affine(ft_delta2[lane * 16 + digit]);
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) {
        if (lane < d) {
            if (wave < 4) sh1[wave][lane] = jac_add_ni(sh1[wave][lane], sh1[wave][lane + d]);
            else sh2[lane] = jac_add_ni(sh2[lane], sh2[lane + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out->fixed_c = jac_add_ni(jac_add_ni(sh1[1][0], sh1[2][0]), sh1[3][0]);
    }
    // the constant terms of a and b join here, off the critical path: k_assemble is left with ONE addition per proof element
    if (threadIdx.x == 64) out->r_delta = jac_madd_ni(sh1[0][0], *alpha1);
    if (threadIdx.x == 256) out->s_delta2 = jac_madd_ni(sh2[0], *beta2);
}
__global__ __launch_bounds__(320) void k_assemble_pre(const G1A* __restrict__ ft_alpha1, const G1A* __restrict__ ft_beta1,
                                                      const G1A* __restrict__ ft_delta1, const G2A* __restrict__ ft_delta2,
                                                      const G1A* __restrict__ alpha1, const G2A* __restrict__ beta2, Fr r, Fr s, AssemblePre* __restrict__ out) {
    ZK_LATENCY_KERNEL();
    assemble_pre_body(ft_alpha1, ft_beta1, ft_delta1, ft_delta2, alpha1, beta2, r, s, out);
}
// batch form (zk_prove_batch_*): workgroup j serves proof j; (r, s) pairs in device memory
__global__ __launch_bounds__(320) void k_assemble_pre_batch(const G1A* __restrict__ ft_alpha1, const G1A* __restrict__ ft_beta1,
                                                            const G1A* __restrict__ ft_delta1, const G2A* __restrict__ ft_delta2,
                                                            const G1A* __restrict__ alpha1, const G2A* __restrict__ beta2, const Fr* __restrict__ rs, AssemblePre* __restrict__ out) {
    ZK_LATENCY_KERNEL();
    assemble_pre_body(ft_alpha1, ft_beta1, ft_delta1, ft_delta2, alpha1, beta2, rs[2 * blockIdx.x], rs[2 * blockIdx.x + 1], out + blockIdx.x);
}

__device__ __forceinline__ void put_be32(const Fq& x_mont, uint8_t* out) {
    Fq x = x_mont.to_canonical();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t w = x.l[7 - i];
        out[4 * i + 0] = (uint8_t)(w >> 24);
        out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8);
        out[4 * i + 3] = (uint8_t)w;
    }
}
// Blinding factors of the three inversions that close a proof (ec.cuh jac_to_affine_vartime): fresh per proof, drawn on the host by
// the library itself -- they do not change the proof bytes, so the (r, s)-determinism of the ABI is untouched.
struct AssembleBlind { Fq a; Fq2 b; Fq c; };

__device__ void encode_g1(const G1J& p, const Fq& lambda, uint8_t* out) {
    for (int i = 0; i < 65; ++i) out[i] = 0;
    if (p.is_inf()) return;
    G1A a = jac_to_affine_vartime(p, lambda);   // one lane per wave is active here
    out[0] = 4;
    put_be32(a.x, out + 1);
    put_be32(a.y, out + 33);
}
__device__ void encode_g2(const G2J& p, const Fq2& lambda, uint8_t* out) {
    for (int i = 0; i < 129; ++i) out[i] = 0;
    if (p.is_inf()) return;
    G2A a = jac_to_affine_vartime(p, lambda);
    out[0] = 4;
    put_be32(a.x.c1, out + 1);
    put_be32(a.x.c0, out + 33);
    put_be32(a.y.c1, out + 65);
    put_be32(a.y.c0, out + 97);
}

// (mod.rs:274-293)  a = A + [alpha + r delta] ;  b = B2 + [beta2 + s delta2] ;   (brackets: k_assemble_pre)
// c = H + L + s a + r (beta + B1 + s delta) - (r s) delta
//   = [H + r B1 + s A] + L + [s alpha + r beta + (r s) delta]
// where H + r B1 + s A comes out of ONE inner product: scalars h_i over xi_t and (r v_i + s u_i) over
// xi.  No scalar multiplication with a run-time base is left.
__device__ __forceinline__ void assemble_body(const MsmResults* __restrict__ ms, const AssemblePre* __restrict__ pre, const AssembleBlind& bl, uint8_t* __restrict__ proof) {
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x & 63) return;
    // the additions in the lazy radix (no conversion per multiplication: 26 us instead of 59 for the Fq2 one, tools/ubench_assemble.hip)
    if (wave == 0) encode_g1(jacr_store(add_lazy(jacr_load(ms->a), jacr_load(pre->r_delta))), bl.a, proof);
    if (wave == 1) encode_g2(jacr_store(add_lazy(jacr_load(ms->b2), jacr_load(pre->s_delta2))), bl.b, proof + 65);
    if (wave == 2) encode_g1(jac_add_ni(jac_add_ni(ms->hb, ms->l), pre->fixed_c), bl.c, proof + 65 + 129);
}
__global__ __launch_bounds__(192) void k_assemble(const MsmResults* __restrict__ ms, const AssemblePre* __restrict__ pre, AssembleBlind bl, uint8_t* __restrict__ proof) {
    ZK_LATENCY_KERNEL();
    assemble_body(ms, pre, bl, proof);
}

// batch form: workgroup j assembles proof j from blob j of the partial sums
__global__ __launch_bounds__(192) void k_assemble_batch(const uint8_t* __restrict__ blobs, const AssemblePre* __restrict__ pre, AssembleBlind bl, uint8_t* __restrict__ proofs) {
    ZK_LATENCY_KERNEL();
    // one draw per batch, made different for every proof of it (bl . (j + 1): still uniform, still unknown)
    const Fq j1 = Fq::from_u32(blockIdx.x + 1);
    bl.a = bl.a * j1; bl.b = Fq2{bl.b.c0 * j1, bl.b.c1 * j1}; bl.c = bl.c * j1;
    assemble_body(reinterpret_cast<const MsmResults*>(blobs + (size_t)blockIdx.x * ZK_PARTIAL_BYTES), pre + blockIdx.x, bl,
                  proofs + (size_t)blockIdx.x * ZK_PROOF_BYTES);
}

__global__ void k_sum_partials(const uint8_t* __restrict__ partials, int world, MsmResults* __restrict__ out) {
    ZK_LATENCY_KERNEL();
    int which = threadIdx.x >> 6;
    if ((threadIdx.x & 63) || which > 4) return;
    if (which < 4) {
        G1J acc = G1J::infinity();
        for (int g = 0; g < world; ++g) {
            const MsmResults* p = reinterpret_cast<const MsmResults*>(partials + (size_t)g * ZK_PARTIAL_BYTES);
            const G1J* src = which == 0 ? &p->a : which == 1 ? &p->hb : which == 2 ? &p->l : &p->spare;
            acc = jac_add_ni(acc, *src);
        }
        G1J* dst = which == 0 ? &out->a : which == 1 ? &out->hb : which == 2 ? &out->l : &out->spare;
        *dst = acc;
    } else {
        G2J acc = G2J::infinity();
        for (int g = 0; g < world; ++g) acc = jac_add_ni(acc, reinterpret_cast<const MsmResults*>(partials + (size_t)g * ZK_PARTIAL_BYTES)->b2);
        out->b2 = acc;
    }
}

static Fr fr_from_words64(const uint64_t w[4]) {
    Fr x;
    for (int i = 0; i < 4; ++i) { x.l[2 * i] = (uint32_t)w[i]; x.l[2 * i + 1] = (uint32_t)(w[i] >> 32); }
    return x;
}

struct AssembleScratch {
    AssemblePre pre;
};

// ---- pipeline state: everything one in-flight proof owns ---------------------------------------
// Several slots let zk_prove_submit enqueue proof k+1 (its sorting / NTT stage / first accumulation)
// while the latency-bound tail of proof k (reductions, assembly, copy-out) is still running.
struct ProveSlot {
    DevBuf<Fr> a_mont, ue, ve, uc_can, vc_can, hb_can, wc, prod_a, prod_b, div_work;
    // roots-of-unity form: pairs of vectors that are transformed together sit back to back (one batched NTT launch per
    // pass for both): uv = V | U evaluations / coefficients, uvg = V | U on the coset, xy = U.V on <w> | on g<w>; each half
    // holds `count` vectors of n elements (count = 1 outside batches)
    DevBuf<Fr> uv, uvg, xy;
    DevBuf<Fr> arb_vals, arb_work;   // form 2 (arbitrary roots): SpMV outputs U | V, scratch of the interpolation
    MsmWorkspace ws[zk_ctx::MSM_STREAMS];
    DevBuf<MsmResults> ms;
    DevBuf<AssembleScratch> as;
    DevBuf<uint8_t> d_proof;
    DevBuf<int> flag;
    uint8_t* h_proof = nullptr;    // pinned
    int* h_flag = nullptr;         // pinned
    // batch form: scalars of the four products for `batch` proofs, their partial sums / (r, s) / proofs
    DevBuf<Fr> bx_l, bx_v, bx_u, bx_h, b_rs;
    DevBuf<uint8_t> b_partials, b_proofs;
    DevBuf<AssemblePre> b_pre;
    DevBuf<Fr> d_wit;              // witness of a proof submitted with a host pointer
    uint8_t* h_b_proofs = nullptr; // pinned, ZK_MAX_BATCH proofs
    Fr* h_b_rs = nullptr;          // pinned
    int batch = 0;
    hipEvent_t fork_evt = nullptr, pre_evt = nullptr, done_evt = nullptr;
    hipEvent_t msm_done[zk_ctx::MSM_STREAMS] = {}, acc_evt[zk_ctx::MSM_STREAMS] = {}, scal_evt[zk_ctx::MSM_STREAMS] = {};
    bool busy = false, partial = false;
    hipStream_t fin_stream = nullptr;   // the stream done_evt was recorded on
    void init() {
        ms.alloc(1); as.alloc(1); d_proof.alloc(ZK_PROOF_BYTES); flag.alloc(1);
        ZK_HIP(hipMemset(ms.p, 0, sizeof(MsmResults)));   // `spare` is never written but travels with the partial sums
        ZK_HIP(hipHostMalloc((void**)&h_proof, ZK_PROOF_BYTES));
        ZK_HIP(hipHostMalloc((void**)&h_flag, sizeof(int)));
        ZK_HIP(hipEventCreateWithFlags(&fork_evt, hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&pre_evt, hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&done_evt, hipEventDisableTiming));
        for (int k = 0; k < zk_ctx::MSM_STREAMS; ++k) {
            ZK_HIP(hipEventCreateWithFlags(&msm_done[k], hipEventDisableTiming));
            ZK_HIP(hipEventCreateWithFlags(&acc_evt[k], hipEventDisableTiming));
            ZK_HIP(hipEventCreateWithFlags(&scal_evt[k], hipEventDisableTiming));
        }
    }
    ~ProveSlot() {
        if (h_proof) (void)hipHostFree(h_proof);
        if (h_flag) (void)hipHostFree(h_flag);
        if (h_b_proofs) (void)hipHostFree(h_b_proofs);
        if (h_b_rs) (void)hipHostFree(h_b_rs);
        for (hipEvent_t e : {fork_evt, pre_evt, done_evt}) if (e) (void)hipEventDestroy(e);
        for (int k = 0; k < zk_ctx::MSM_STREAMS; ++k) {
            if (msm_done[k]) (void)hipEventDestroy(msm_done[k]);
            if (acc_evt[k]) (void)hipEventDestroy(acc_evt[k]);
            if (scal_evt[k]) (void)hipEventDestroy(scal_evt[k]);
        }
    }
};
struct ProveState {
    static constexpr int SLOTS = ZK_MAX_IN_FLIGHT;
    // Blinding factors only (AssembleBlind): nothing drawn here reaches the proof bytes, but lambda hides the witness-dependent Z from
    // the trip count of the closing inversions, so it has to be unpredictable: a ChaCha20 key stream under 256 bits of OS entropy
    // (getrandom), not a 32-bit-seeded Mersenne twister.
    struct BlindRng {
        uint32_t key[8], block[16];
        uint64_t counter = 0;
        int used = 16;
        BlindRng() {
            size_t got = 0;
            while (got < sizeof(key)) {
                const ssize_t r = getrandom(reinterpret_cast<uint8_t*>(key) + got, sizeof(key) - got, 0);
                if (r <= 0) break;
                got += (size_t)r;
            }
            if (got < sizeof(key)) {   // no getrandom (very old kernel): every word from the C++ entropy source
                std::random_device rd;
                for (uint32_t& w : key) w = rd();
            }
        }
        static uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
        void refill() {
            uint32_t x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                              (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
            uint32_t w[16];
            std::memcpy(w, x, sizeof(w));
            auto qr = [&](int a, int b, int c, int d) {
                w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12);
                w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
            };
            for (int round = 0; round < 10; ++round) {
                qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
                qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
            }
            for (int i = 0; i < 16; ++i) block[i] = w[i] + x[i];
            ++counter;
            used = 0;
        }
        uint64_t operator()() {
            if (used > 14) refill();
            const uint64_t v = (uint64_t)block[used] | ((uint64_t)block[used + 1] << 32);
            used += 2;
            return v;
        }
    } rng;
    AssembleBlind draw_blind() {
        auto fq = [&] {
            Fq x;
            for (int i = 0; i < 4; ++i) { const uint64_t w = rng(); x.l[2 * i] = (uint32_t)w; x.l[2 * i + 1] = (uint32_t)(w >> 32); }
            x.l[7] &= 0x0fffffffu;   // < 2^252 < q: a valid residue whatever form it is read in
            x.l[0] |= 1u;            // never zero
            return x;
        };
        AssembleBlind b;
        b.a = fq(); b.b = Fq2{fq(), fq()}; b.c = fq();
        return b;
    }
    ProveSlot slot[SLOTS];
    int next = 0;
    hipEvent_t last_acc = nullptr;   // end of the most recently enqueued accumulation chain
    // zk_prove_combine scratch (allocated once)
    DevBuf<MsmResults> comb_ms;
    DevBuf<AssembleScratch> comb_as;
    DevBuf<uint8_t> comb_proof;
    uint8_t* comb_h_proof = nullptr;
    ProveState() {
        for (auto& s : slot) s.init();
        comb_ms.alloc(1); comb_as.alloc(1); comb_proof.alloc(ZK_PROOF_BYTES);
        ZK_HIP(hipHostMalloc((void**)&comb_h_proof, ZK_PROOF_BYTES));
    }
    ~ProveState() { if (comb_h_proof) (void)hipHostFree(comb_h_proof); }
};

// consecutive tickets alternate between two main streams, so that the SpMV / NTT stage of one proof does not queue
// behind the (contended, ~25 kernel) stage of the previous one; every helper launches on ctx->stream
struct StreamSwap {
    zk_ctx* c; hipStream_t saved;
    StreamSwap(zk_ctx* c_, hipStream_t s) : c(c_), saved(c_->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = saved; c->cur_slot = -1; }
};

// Stream of inner product k (0 = B in G2, 1 = L, 2 = A, 4 = H + r B1 + s A): its sort, its accumulation and its reduction tail
static hipStream_t msm_stream_for(zk_ctx* ctx, int k) { return ctx->msm_stream[k]; }

// The next submission takes the LOWEST free slot (not the next one in a ring): a caller that keeps d proofs in flight then only
// ever touches d slots -- their buffers (2.5 GiB each at 2^20, allocated on first use) are allocated during the first d proofs
// and never inside a steady-state region, and with two in flight consecutive proofs still alternate between slots 0 and 1 (and so
// between the two main streams).
static void pick_next_slot(ProveState& ps) {
    for (int k = 0; k < ProveState::SLOTS; ++k)
        if (!ps.slot[k].busy) { ps.next = k; return; }
    ps.next = (ps.next + 1) % ProveState::SLOTS;   // all busy: the next submit reports it
}

static ProveState& prove_state(zk_ctx* ctx) {
    if (!ctx->prove_state) ctx->prove_state = std::make_shared<ProveState>();
    return *ctx->prove_state;
}

// ---- scalar exchange (multi-GPU) ---------------------------------------------------
// Rank g owns the points [g c, (g+1) c) of every inner product, c = ceil(count / world).  The owner of a proof
// computes its scalars once (prove_submit with xout) into `world` chunks of c scalars per product; after an
// all-to-all every rank holds, for each proof of the round, the chunk that multiplies its own points.
struct ExchangeDims { size_t cl, cn, ch; };
static ExchangeDims exchange_dims(const zk_qap& q, int world) {
    const size_t w = (size_t)world, nl = q.m > q.input + 1 ? q.m - q.input - 1 : 0;
    return ExchangeDims{std::max<size_t>((nl + w - 1) / w, 1), (q.n + w - 1) / w, (2 * q.n + w - 1) / w};   // never an empty array
}
void prove_exchange_elems(const zk_qap& q, int world, size_t out[4]) {
    const ExchangeDims xd = exchange_dims(q, world);
    out[0] = xd.cl * world; out[1] = out[2] = xd.cn * world; out[3] = xd.ch * world;
}

static void launch_pre(zk_ctx* ctx, const zk_crs& crs, hipStream_t st, const Fr& rc, const Fr& sc, AssembleScratch* d_as) {
    hipLaunchKernelGGL(k_assemble_pre, dim3(1), dim3(320), 0, st, crs.ft_alpha1.p, crs.ft_beta1.p, crs.ft_delta1.p, crs.ft_delta2.p, crs.alpha1.p, crs.beta2.p, rc, sc, &d_as->pre);
    ZK_HIP(hipGetLastError());
}

// The form a proof over (crs, q) takes, decided in ONE place for every entry point (a single proof, a batch, the scalars and the
// inner products of the multi-GPU exchange -- ADVICE r3: they used to differ):
//   0 roots of unity, 1 integer roots in the evaluation basis, 2 coefficients by the sub-product tree (arbroots.hip), 3 dense.
// An integer-roots QAP over a CRS that carries only the powers (zk_crs_upload, ZKCRSv1) gets the Lagrange-basis points by the change
// of basis, once per CRS: the transposed interpolation tree (gbasis.hip, O(n log^2 n) point operations, to 2^22 gates) or, below
// basis_tree_min gates, the n^2 inner products (basis.hip, to basis_max_n()); with the tree switched off it proves beyond that in
// form 2 with the roots 1..n as caller data -- the same bytes from ANY CRS.  One-off tables of forms 2 / 3 (the power-series inverse of rev(t)) are built here, before anything is enqueued.
static int prove_form(zk_ctx* ctx, zk_crs& crs, const zk_qap& q) {
    int form = q.dense ? 3 : q.roots;
    if (form == 1 && !crs.ap) {
        const bool tree = ctx->opt_basis_tree_min >= 0 && q.n >= (size_t)ctx->opt_basis_tree_min && q.n <= ((size_t)1 << (NTT_MAX_LOG - 2));
        if (tree || q.n <= basis_max_n()) crs_lagrange_from_powers(ctx, crs, q);
        else { arb_attach_integer_roots(ctx, const_cast<zk_qap&>(q)); form = 2; }
    }
    if (form >= 2 && !q.t_is_zero && 2 * q.n - 1 > q.t_degree && 2 * q.n - 1 - q.t_degree >= 512 && !ctx->opt_long_division) {
        unsigned lc0 = 1;
        while (((size_t)1 << lc0) < 2 * q.n) ++lc0;
        qap_ensure_tinv(ctx, const_cast<zk_qap&>(q), 2 * q.n - 1 - q.t_degree, lc0);
    }
    return form;
}

// Form 2, one proof: the SpMV outputs (values of U, V on the caller's roots) are interpolated by the sub-product tree, A and B take the
// coefficients, h is the quotient of U V by t.  vc_can / uc_can: n canonical scalars each; hb_can: h (n - 1) | r v + s u (n).
// `uv_ready` is called when uc_can and vc_can exist (their inner products can start while the quotient is computed).
template <class Ready>
static void arb_scalar_stage(zk_ctx* ctx, ProveSlot& S, const zk_qap& q, const Fr* a_can, size_t a_len, const Fr& r_mont, const Fr& s_mont,
                             Fr* vc_can, Fr* uc_can, Fr* hb_can, Ready&& uv_ready) {
    const size_t n = q.n;
    hipStream_t st = ctx->stream;
    unsigned lc = 1;
    while (((size_t)1 << lc) < 2 * n) ++lc;
    const size_t nc = (size_t)1 << lc;
    S.ue.ensure(n); S.ve.ensure(n); S.prod_a.ensure(nc); S.prod_b.ensure(nc);
    S.arb_vals.ensure(2 * n); S.arb_work.ensure(arb_work_elems(q));
    // W is not needed: U V = h t + E with E = the interpolant of U_k V_k (degree < n), so the quotient of U V alone is h.
    spmv(ctx, q.u_gate, a_can, a_len, S.arb_vals.p);         // the witness as given (canonical: k_spmv)
    spmv(ctx, q.v_gate, a_can, a_len, S.arb_vals.p + n);
    arb_coefficients(ctx, q, S.arb_vals.p, S.arb_work.p, S.ue.p, S.ve.p);
    fr_from_mont(ctx, S.ue.p, uc_can, n);
    fr_from_mont(ctx, S.ve.p, vc_can, n);
    uv_ready();
    fr_lincomb_to_canonical(ctx, S.ve.p, r_mont, S.ue.p, s_mont, hb_can + (n - 1), n);   // bases: xi_t (n-1) | xi (n)
    ZK_HIP(hipMemsetAsync(S.prod_a.p, 0, nc * sizeof(Fr), st));
    ZK_HIP(hipMemsetAsync(S.prod_b.p, 0, nc * sizeof(Fr), st));
    ZK_HIP(hipMemcpyAsync(S.prod_a.p, S.ue.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ZK_HIP(hipMemcpyAsync(S.prod_b.p, S.ve.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ntt_dif(ctx, S.prod_a.p, lc, false, false);
    ntt_dif(ctx, S.prod_b.p, lc, false, false);
    fr_pointwise_mul(ctx, S.prod_a.p, S.prod_b.p, S.prod_a.p, nc);
    ntt_dit(ctx, S.prod_a.p, lc, true, true, nullptr);                // U*V coefficients, natural order
    ZK_HIP(hipMemsetAsync(S.prod_b.p, 0, nc * sizeof(Fr), st));
    const size_t len_r = 2 * n - 1, d = q.t_degree;                   // t is monic of degree n: n - 1 quotient coefficients
    if (len_r > d) {
        if (len_r - d >= 512 && !ctx->opt_long_division) {
            S.div_work.ensure(nc);
            poly_divide_newton(ctx, q, S.prod_a.p, len_r, lc, S.div_work.p, S.prod_b.p);
        } else {
            poly_divide(ctx, S.prod_a.p, len_r, q.dt.p, d, q.t_cinv.p, S.prod_b.p);
        }
    }
    if (n > 1) fr_from_mont(ctx, S.prod_b.p, hb_can, n - 1);
}

// SpMV / NTT stage of one proof in the roots-of-unity form: the scalars of the inner products B2 (vc), A (uc) and
// H + r B1 + s A (hb: h | r v + s u) from the witness as the caller gave it (canonical; range-checked by the caller).  `launch(k, after, scalars, count)`
// is called as soon as the scalars of product k exist (k = MSM stream: 1 L, 0 B2, 2 A, 4 HB).
// accumulation chain L -> B2 -> A -> H+rB1+sA: L needs only the witness, so the chip is busy
// ~0.6 ms after the call starts; the long G2 reduction tail hides behind A and the H product
template <class Launch>
static void sparse_scalar_stage(zk_ctx* ctx, ProveSlot& S, const zk_qap& q, NttTables& tabs_ref, const Fr* d_weights, size_t a_len, size_t n_l,
                                const Fr& r_mont, const Fr& s_mont, Fr* vc_can, Fr* uc_can, Fr* hb_can, Launch&& launch) {
    const size_t n = q.n, l = q.input;
    NttTables* tabs = &tabs_ref;
    hipStream_t st = ctx->stream;
    Fr *ve = S.uv.p, *ue = S.uv.p + n, *x0 = S.xy.p, *y0 = S.xy.p + n, *vg = S.uvg.p, *ug = S.uvg.p + n;
    launch(1, -1, d_weights + l + 1, n_l);
    spmv(ctx, q.u_gate, d_weights, a_len, ue);
    spmv(ctx, q.v_gate, d_weights, a_len, ve);
    // The inverse transforms run WITHOUT their 1 / n (a multiplication per element in the last pass): the factor rides in the
    // kernels that consume their outputs anyway -- the conversions to canonical scalars, r v + s u, the coset table, the h combine.
    const Fr n_inv = tabs->n_inv;
    if (ctx->opt_ntt_fuse && ntt_dif_fusable(q.log_n)) {
        // Two-pass sizes: the element-wise kernels around the three DIF transforms ride in their tile loads / stores (ntt_dif_fused):
        // the first transform reads the SpMV outputs and writes uvg (the evaluations in uv stay for U.V on <w>), its last pass also
        // emits the canonical scalars of A and B; U.V on <w> and on g<w> are formed in the load of the last transform.  No
        // k_pointwise_mul, no k_scale_to_canonical, no copy uv -> uvg.
        NttFuse f1;
        f1.src_a[0] = ve; f1.src_a[1] = ue; f1.half = 1;
        f1.canon_out[0] = vc_can; f1.canon_out[1] = uc_can; f1.canon_k = n_inv;
        ntt_dif_fused(ctx, S.uvg.p, q.log_n, true, 2, f1);             // uvg = n V | n U coefficients (bit-reversed order)
        launch(0, 1, vc_can, n);
        launch(2, 0, uc_can, n);
        fr_lincomb_to_canonical(ctx, vg, r_mont * n_inv, ug, s_mont * n_inv, hb_can + n, n);
        ntt_dit(ctx, S.uvg.p, q.log_n, false, false, tabs->coset_fwd_brev.p, 2);   // V, U on g<w> (the table holds g^i / n)
        NttFuse f3;
        f3.src_a[0] = ue; f3.src_b[0] = ve; f3.src_a[1] = ug; f3.src_b[1] = vg; f3.half = 1;
        ntt_dif_fused(ctx, S.xy.p, q.log_n, true, 2, f3);              // n (lo + hi) | n (lo - hi)_i * g^i
    } else {
    fr_pointwise_mul(ctx, ue, ve, x0, n);                             // U.V on <w>
    ntt_dif(ctx, S.uv.p, q.log_n, true, false, 2);                    // n V, n U coefficients (bit-reversed order), one launch per pass
    fr_scale_to_canonical(ctx, ve, n_inv, vc_can, n);
    launch(0, 1, vc_can, n);
    fr_scale_to_canonical(ctx, ue, n_inv, uc_can, n);
    launch(2, 0, uc_can, n);
    // r v_i + s u_i: B in G1 (needed only as r*B1) and s*A are folded into the H product as scalars
    fr_lincomb_to_canonical(ctx, ve, r_mont * n_inv, ue, s_mont * n_inv, hb_can + n, n);
    ZK_HIP(hipMemcpyAsync(S.uvg.p, S.uv.p, 2 * n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ntt_dit(ctx, S.uvg.p, q.log_n, false, false, tabs->coset_fwd_brev.p, 2);   // V, U on g<w> (the table holds g^i / n)
    fr_pointwise_mul(ctx, ug, vg, y0, n);                             // U.V on g<w>
    ntt_dif(ctx, S.xy.p, q.log_n, true, false, 2);                    // n (lo + hi) | n (lo - hi)_i * g^i
    }
    Fr half = host_fr_from_u64(2).inv() * n_inv;
    h_combine(ctx, x0, y0, tabs->coset_inv_brev_half.p, half, hb_can, n);        // the table holds g^-i / (2 n)
    // bases: xi_t (n entries, entry brev(n-1) = n-1 is infinity) | xi (n entries)
    launch(4, 2, hb_can, 2 * n);
}

// Enqueues one proof (or one rank's partial sums) and returns without waiting.  ticket = slot index.
// xout != nullptr: "scalars only" -- the SpMV / NTT stage of one proof, no inner products; the scalars of the four
// products are written to xout[0..3] = L | V | U | H,k in the exchange layout of prove_exchange_elems(world).
int prove_submit(zk_ctx* ctx, const zk_crs& crs_c, const zk_qap& qap_c, const Fr* d_weights, size_t m_in, const uint64_t* r, const uint64_t* s,
                 int rank, int world, void* d_partial_out, Fr* const* xout) {
    zk_crs& crs = const_cast<zk_crs&>(crs_c);   // lazily built tables only
    const zk_qap& q = qap_c;
    ZK_REQUIRE(crs.n == q.n && crs.m == q.m && crs.input == q.input, ZK_ERR_ARG, "prove: CRS and QAP dimensions differ");
    ZK_REQUIRE(d_partial_out || world == 1 || xout, ZK_ERR_ARG, "prove: world > 1 needs a partial output buffer");
    ZK_REQUIRE(!xout || !qap_c.dense, ZK_ERR_UNSUPPORTED, "prove: the scalar exchange needs a sparse QAP form");
    Fr rc = fr_from_words64(r), sc = fr_from_words64(s);
    ZK_REQUIRE(rc.raw_in_range() && sc.raw_in_range(), ZK_ERR_RANGE, "prove: r or s >= modulus");
    ZK_REQUIRE(!q.dense || !q.t_is_zero, ZK_ERR_DIV_BY_ZERO, "Dividend must be non-zero");   // field/mod.rs:440
    const Fr r_mont = Fr::from_canonical(rc), s_mont = Fr::from_canonical(sc);
    ProveState& ps = prove_state(ctx);
    const int ticket = ps.next;
    ProveSlot& S = ps.slot[ticket];
    ZK_REQUIRE(!S.busy, ZK_ERR_ARG, "prove: too many proofs in flight (call zk_prove_wait first)");
    // one-off table construction happens before anything of this proof is enqueued
    // an integer-roots QAP over a CRS that carries only the powers (zk_crs_upload, ZKCRSv1): change of basis, once per CRS
    int form = prove_form(ctx, crs, q);
    ZK_REQUIRE(!(form == 3 && xout), ZK_ERR_UNSUPPORTED, "prove: the scalar exchange needs a sparse QAP form");
    if (xout) {}   // scalars only: no inner product, no table (the ranks of a scalar exchange build only their own slices)
    else if (form >= 2) crs_ensure_tables(ctx, crs, false, 0);   // coefficient forms: the reference's [x^i], natural order
    else if (form == 1) crs_ensure_tables(ctx, crs, false, 0, true);
    else crs_ensure_tables(ctx, crs, true, q.log_n);
    if (!d_partial_out && !xout) crs_ensure_fixed_tables(ctx, crs);

    const size_t n = q.n, m = q.m, l = q.input;
    const size_t a_len = std::min(m_in, m);   // zip(weights) truncates (mod.rs:233-253)
    // consecutive proofs alternate between two main streams, so that the SpMV / NTT stage of proof k+1 does not
    // queue behind the (contended, ~25 kernel) stage of proof k; every helper launches on ctx->stream
    StreamSwap swap_guard(ctx, (ticket & 1) ? ctx->main_alt : ctx->stream);
    hipStream_t st = ctx->stream;
    ctx->cur_slot = ticket;
    S.batch = 0;
    S.partial = d_partial_out != nullptr || xout != nullptr;

    if (ctx->submit_wait_evt) { ZK_HIP(hipStreamWaitEvent(st, ctx->submit_wait_evt, 0)); ctx->submit_wait_evt = nullptr; }
    ZK_HIP(hipMemsetAsync(S.flag.p, 0, sizeof(int), st));
    // Sparse forms multiply the witness as the caller gave it (canonical integers; the by-gate coefficients carry the extra R, qap.hip
    // k_spmv): only its range is checked.  The dense form still converts.
    if (qap_c.dense) {
        S.a_mont.ensure(std::max<size_t>(a_len, 1));
        fr_to_mont(ctx, d_weights, S.a_mont.p, a_len, S.flag.p);
    } else {
        fr_check_range(ctx, d_weights, a_len, S.flag.p);
    }

    // the r/s-only fixed-base multiplications run on the side stream beside everything below
    if (!d_partial_out && !xout) {
        hipStream_t pre_st = ctx->opt_serialize ? st : ctx->side;
        launch_pre(ctx, crs, pre_st, rc, sc, S.as.p);
        ZK_HIP(hipEventRecord(S.pre_evt, pre_st));
    }

    // The inner products run on their own streams, each forked from the main stream as soon as its
    // scalars exist.  `after`: MSM slot whose accumulation must finish first (-1: the previous proof's
    // chain).  The accumulation kernels each fill every SIMD, so they are chained in a chosen order
    // instead of thrashing each other; sorting phases and reduction tails overlap freely.
    MsmResults* ms = S.ms.p;
    const size_t n_l = a_len > l + 1 ? std::min(a_len - l - 1, m - l - 1) : 0;
    // `off`: first point of the table the scalars multiply (L on its own sits behind xi_t | xi in t_hb1); `sp`: the merged product's
    // second scalar array (MsmSplit).
    auto launch_now = [&](int k, int after, auto& table, const Fr* scalars, size_t count, auto* out, hipStream_t ms_st, size_t off, MsmSplit sp) {
        ZK_HIP(hipStreamWaitEvent(ms_st, S.scal_evt[k], 0));
        hipEvent_t wait_evt = after >= 0 ? S.acc_evt[after] : ps.last_acc;
        hipStream_t end_st;
        if (world > 1 && ctx->opt_shard_points == 1) {
            // partial sums by point ranges: rank g takes the scalars / bases [count g / world, count (g+1) / world) with every window
            const size_t lo = count * (size_t)rank / (size_t)world, hi = count * ((size_t)rank + 1) / (size_t)world;
            end_st = msm_run(ctx, S.ws[k], ms_st, table, scalars + lo, hi - lo, 0, 1, out, wait_evt, S.acc_evt[k], off + lo);
        } else {
            MsmGroups ex;
            ex.bucket_shard = world > 1 && ctx->opt_shard_points == 2;   // partial sums by bucket range (1 / world of the entries AND of the buckets)
            // measurement switch tail_stream: the merged product's reduction tail on the idle L stream, so that the next proof's sort of
            // the same product starts when this accumulation ends instead of ~1 ms later (r5_experiments.txt item 11)
            if (k == 4 && sp.scalars2 && ctx->opt_tail_stream && !ctx->opt_serialize) ex.tail_stream = ctx->msm_stream[1];
            end_st = msm_run(ctx, S.ws[k], ms_st, table, scalars, count, rank, world, out, wait_evt, S.acc_evt[k], off, ex, sp);
        }
        ZK_HIP(hipEventRecord(S.msm_done[k], end_st));
        ps.last_acc = S.acc_evt[k];
    };
    // L and H + r B1 + s A only ever occur ADDED in the proof element c (assemble_body), so they are one inner product over the table
    // xi_t | xi | sum_delta with the scalars h | r v + s u (the slot's array) and the witness (the caller's): one set of 2^(c-1)
    // buckets instead of two, i.e. one reduction tail (2 additions per bucket), one sort, ~25 launches less per proof (option merge_lh;
    // round 5: profiles/r5_experiments.txt).  ms->l stays infinity.  Not with partial sums by point ranges (they slice ONE scalar array).
    const bool merge_lh = ctx->opt_merge_lh && !xout && !(world > 1 && ctx->opt_shard_points == 1);
    const Fr* l_scalars = nullptr;
    size_t l_count = 0;
    bool l_pending = false;
    // The ~30 launches of an inner product are enqueued AFTER the whole SpMV / NTT stage (an event marks the point of the main
    // stream where its scalars exist): a lone proof of a small circuit is bound by the host's enqueue rate, and with the
    // products enqueued in between the stage's own 25 short kernels sat 2 ms apart on the timeline of a 2^16 proof.
    // They are enqueued in the order of the accumulation chain (option chain_order): every accumulation waits for the one before it.
    std::vector<std::pair<int, std::function<void(int)>>> deferred;
    auto launch = [&](int k, int, auto& table, const Fr* scalars, size_t count, auto* out) {
        if (xout) return;
        size_t off = 0;
        MsmSplit sp;
        if constexpr (std::is_same<std::remove_reference_t<decltype(table)>, MsmTable<Fq>>::value) {
            if (k == 1) {
                if (merge_lh) { l_scalars = scalars; l_count = count; l_pending = true; return; }   // joins product 4
                off = crs.off_l;
            } else if (k == 4 && l_pending && count == crs.off_l) {
                sp.scalars2 = l_scalars; sp.split = crs.off_l; sp.n2 = l_count;
                count += l_count;
                l_pending = false;
                ZK_HIP(hipMemsetAsync(&ms->l, 0, sizeof(G1J), st));   // infinity: everything is in hb
            }
        }
        // Measurement switch alt_stream (default 0): the merged product of odd tickets on the (now idle) L stream.  On ONE stream its
        // sort queues behind the previous proof's reduction tail and every other proof's last accumulation starts ~0.9 ms late
        // (profiles/r5_timeline_pipelined_2p20.txt) -- but with the sort moved forward the accumulation it then overlaps runs 1.2 ms
        // longer: 103.9 (one stream) against 102.4 proofs/s.  The chip is bound by the work, not by that gap (r5_experiments.txt item 8).
        const int sk = (k == 4 && sp.scalars2 && (ticket & 1) && ctx->opt_alt_stream) ? 1 : k;
        hipStream_t ms_st = ctx->opt_serialize ? st : msm_stream_for(ctx, sk);   // serialize: measurement mode, no overlap at all
        ZK_HIP(hipEventRecord(S.scal_evt[k], st));
        auto* tab = &table;
        deferred.emplace_back(k, [&, k, tab, scalars, count, out, ms_st, off, sp](int after) { launch_now(k, after, *tab, scalars, count, out, ms_st, off, sp); });
    };
    auto run_deferred = [&] {
        static const int orders[3][4] = {{1, 0, 2, 4}, {1, 2, 0, 4}, {0, 1, 4, 2}};   // MSM slots: 1 = L, 0 = B in G2, 2 = A, 4 = H + r B1 + s A
        const int* order = orders[ctx->opt_chain_order == 1 ? 1 : ctx->opt_chain_order == 2 ? 2 : 0];
        int prev = -1;
        for (int pos = 0; pos < 4; ++pos)
            for (auto& d : deferred)
                if (d.first == order[pos]) { d.second(prev); prev = d.first; }
    };
    if (form == 1) {
        // integer roots 1..n (aproots.hip): everything stays in the evaluation basis; bases = Lagrange-basis points
        const size_t M = (size_t)1 << q.ap->log_m;
        S.uv.ensure(2 * n); S.xy.ensure(3 * M);
        Fr *vc_can, *uc_can, *hb_can;
        if (xout) {
            // exchange layout (as below): `world` equal chunks per product, zero scalars behind the last point
            const ExchangeDims xd = exchange_dims(q, world);
            vc_can = xout[1]; uc_can = xout[2]; hb_can = xout[3];
            ZK_HIP(hipMemsetAsync(xout[0], 0, xd.cl * world * sizeof(Fr), st));
            if (n_l) ZK_HIP(hipMemcpyAsync(xout[0], d_weights + l + 1, n_l * sizeof(Fr), hipMemcpyDeviceToDevice, st));
            if (xd.cn * world > n) {
                ZK_HIP(hipMemsetAsync(vc_can + n, 0, (xd.cn * world - n) * sizeof(Fr), st));
                ZK_HIP(hipMemsetAsync(uc_can + n, 0, (xd.cn * world - n) * sizeof(Fr), st));
            }
            ZK_HIP(hipMemsetAsync(hb_can + (2 * n - 1), 0, (xd.ch * world - (2 * n - 1)) * sizeof(Fr), st));
        } else {
            S.uc_can.ensure(n); S.vc_can.ensure(n); S.hb_can.ensure(2 * n);
            vc_can = S.vc_can.p; uc_can = S.uc_can.p; hb_can = S.hb_can.p;
        }
        Fr *ve = S.uv.p, *ue = S.uv.p + n;
        launch(1, -1, crs.t_hb1, d_weights + l + 1, n_l, &ms->l);
        spmv(ctx, q.u_gate, d_weights, a_len, ue);
        spmv(ctx, q.v_gate, d_weights, a_len, ve);
        fr_from_mont(ctx, ve, vc_can, n);
        launch(0, 1, crs.t_xi2, vc_can, n, &ms->b2);                     // B = sum V_k [L_k(x)]_2
        fr_from_mont(ctx, ue, uc_can, n);
        launch(2, 0, crs.t_xi1, uc_can, n, &ms->a);                      // A = sum U_k [L_k(x)]_1
        fr_lincomb_to_canonical(ctx, ve, r_mont, ue, s_mont, hb_can + (n - 1), n);   // bases: L^S t/delta (n-1) | L (n)
        ap_quotient_values(ctx, q, ue, ve, S.xy.p, hb_can);              // h on S = {n+1 .. 2n-1}
        launch(4, 2, crs.t_hb1, hb_can, 2 * n - 1, &ms->hb);
    } else if (form == 0) {
        auto tabs = ntt_get_tables(ctx, q.log_n);
        ntt_ensure_coset_tables(ctx, *tabs);
        S.uv.ensure(2 * n); S.uvg.ensure(2 * n); S.xy.ensure(2 * n);
        Fr *vc_can, *uc_can, *hb_can;
        if (xout) {
            // exchange layout: `world` equal chunks per product, zero scalars behind the last point
            const ExchangeDims xd = exchange_dims(q, world);
            vc_can = xout[1]; uc_can = xout[2]; hb_can = xout[3];
            ZK_HIP(hipMemsetAsync(xout[0], 0, xd.cl * world * sizeof(Fr), st));
            if (n_l) ZK_HIP(hipMemcpyAsync(xout[0], d_weights + l + 1, n_l * sizeof(Fr), hipMemcpyDeviceToDevice, st));
            if (xd.cn * world > n) {
                ZK_HIP(hipMemsetAsync(vc_can + n, 0, (xd.cn * world - n) * sizeof(Fr), st));
                ZK_HIP(hipMemsetAsync(uc_can + n, 0, (xd.cn * world - n) * sizeof(Fr), st));
            }
            if (xd.ch * world > 2 * n) ZK_HIP(hipMemsetAsync(hb_can + 2 * n, 0, (xd.ch * world - 2 * n) * sizeof(Fr), st));
        } else {
            S.uc_can.ensure(n); S.vc_can.ensure(n); S.hb_can.ensure(2 * n);
            vc_can = S.vc_can.p; uc_can = S.uc_can.p; hb_can = S.hb_can.p;
        }
        sparse_scalar_stage(ctx, S, q, *tabs, d_weights, a_len, n_l, r_mont, s_mont, vc_can, uc_can, hb_can,
                            [&](int k, int after, const Fr* scalars, size_t count) {
                                if (k == 1) launch(1, after, crs.t_hb1, scalars, count, &ms->l);           // L: sum a_i * sum_delta_i (merged into HB: launch)
                                else if (k == 0) launch(0, after, crs.t_xi2, scalars, count, &ms->b2);     // B in G2
                                else if (k == 2) launch(2, after, crs.t_xi1, scalars, count, &ms->a);      // A
                                else launch(4, after, crs.t_hb1, scalars, count, &ms->hb);                // H + r B1 + s A: last in the chain
                            });
    } else if (form == 2) {
        // sparse rows over the caller's roots (arbroots.hip), or an integer-roots QAP over a powers-only CRS beyond the change of basis
        Fr *vc_can, *uc_can, *hb_can;
        if (xout) {
            // exchange layout: `world` equal chunks per product, zero scalars behind the last point
            const ExchangeDims xd = exchange_dims(q, world);
            vc_can = xout[1]; uc_can = xout[2]; hb_can = xout[3];
            ZK_HIP(hipMemsetAsync(xout[0], 0, xd.cl * world * sizeof(Fr), st));
            if (n_l) ZK_HIP(hipMemcpyAsync(xout[0], d_weights + l + 1, n_l * sizeof(Fr), hipMemcpyDeviceToDevice, st));
            if (xd.cn * world > n) {
                ZK_HIP(hipMemsetAsync(vc_can + n, 0, (xd.cn * world - n) * sizeof(Fr), st));
                ZK_HIP(hipMemsetAsync(uc_can + n, 0, (xd.cn * world - n) * sizeof(Fr), st));
            }
            ZK_HIP(hipMemsetAsync(hb_can + (2 * n - 1), 0, (xd.ch * world - (2 * n - 1)) * sizeof(Fr), st));
        } else {
            S.uc_can.ensure(n); S.vc_can.ensure(n); S.hb_can.ensure(2 * n);
            vc_can = S.vc_can.p; uc_can = S.uc_can.p; hb_can = S.hb_can.p;
        }
        launch(1, -1, crs.t_hb1, d_weights + l + 1, n_l, &ms->l);
        arb_scalar_stage(ctx, S, q, d_weights, a_len, r_mont, s_mont, vc_can, uc_can, hb_can, [&] {
            launch(2, 1, crs.t_xi1, uc_can, n, &ms->a);
            launch(0, 2, crs.t_xi2, vc_can, n, &ms->b2);
        });
        launch(4, 0, crs.t_hb1, hb_can, 2 * n - 1, &ms->hb);
    } else {
        // dense form: the literal coefficient matrices of QAP<CoefficientPoly>
        unsigned lc = 1;
        while (((size_t)1 << lc) < 2 * n) ++lc;
        size_t nc = (size_t)1 << lc;
        S.ue.ensure(n); S.ve.ensure(n); S.wc.ensure(n); S.prod_a.ensure(nc); S.prod_b.ensure(nc);
        S.uc_can.ensure(n); S.vc_can.ensure(n); S.hb_can.ensure(2 * n);
        launch(1, -1, crs.t_hb1, d_weights + l + 1, n_l, &ms->l);
        dense_matvec(ctx, q.du.p, S.a_mont.p, a_len, n, S.ue.p);
        dense_matvec(ctx, q.dv.p, S.a_mont.p, a_len, n, S.ve.p);
        dense_matvec(ctx, q.dw.p, S.a_mont.p, a_len, n, S.wc.p);
        fr_from_mont(ctx, S.ue.p, S.uc_can.p, n);
        fr_from_mont(ctx, S.ve.p, S.vc_can.p, n);
        launch(2, 1, crs.t_xi1, S.uc_can.p, n, &ms->a);
        launch(0, 2, crs.t_xi2, S.vc_can.p, n, &ms->b2);
        fr_lincomb_to_canonical(ctx, S.ve.p, r_mont, S.ue.p, s_mont, S.hb_can.p + (n - 1), n);   // bases: xi_t (n-1) | xi (n)
        ZK_HIP(hipMemsetAsync(S.prod_a.p, 0, nc * sizeof(Fr), st));
        ZK_HIP(hipMemsetAsync(S.prod_b.p, 0, nc * sizeof(Fr), st));
        ZK_HIP(hipMemcpyAsync(S.prod_a.p, S.ue.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipMemcpyAsync(S.prod_b.p, S.ve.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ntt_dif(ctx, S.prod_a.p, lc, false, false);
        ntt_dif(ctx, S.prod_b.p, lc, false, false);
        fr_pointwise_mul(ctx, S.prod_a.p, S.prod_b.p, S.prod_a.p, nc);
        ntt_dit(ctx, S.prod_a.p, lc, true, true, nullptr);                // U*V coefficients, natural order
        fr_sub_inplace(ctx, S.prod_a.p, S.wc.p, n);                       // - W
        // quotient by t (degree d); remainder dropped (coefficient_poly.rs:148-157)
        ZK_HIP(hipMemsetAsync(S.prod_b.p, 0, nc * sizeof(Fr), st));
        size_t len_r = 2 * n - 1, d = q.t_degree;
        if (len_r > d) {
            // long division below 512 quotient coefficients (as the reference); above, the O(n log n) form
            if (len_r - d >= 512 && !ctx->opt_long_division) {
                S.div_work.ensure(nc);
                poly_divide_newton(ctx, q, S.prod_a.p, len_r, lc, S.div_work.p, S.prod_b.p);
            } else {
                poly_divide(ctx, S.prod_a.p, len_r, q.dt.p, d, q.t_cinv.p, S.prod_b.p);
            }
        }
        fr_from_mont(ctx, S.prod_b.p, S.hb_can.p, n - 1);
        launch(4, 0, crs.t_hb1, S.hb_can.p, 2 * n - 1, &ms->hb);
    }

    ZK_REQUIRE(!l_pending, ZK_ERR_ARG, "prove: internal -- the witness product was not merged");
    run_deferred();
    // join + assembly + copy-out on the finish stream, so that the main stream is free for the next proof.  A
    // scalars-only ticket completes on its own main stream: the finish stream may hold the join of an earlier ticket's
    // inner products, which would delay this one's completion by a whole round.
    hipStream_t fin = xout ? st : ctx->finish;
    if (!xout) {
        ZK_HIP(hipEventRecord(S.fork_evt, st));
        ZK_HIP(hipStreamWaitEvent(fin, S.fork_evt, 0));
        for (int k = 0; k < zk_ctx::MSM_STREAMS; ++k)
            if (k != 3) ZK_HIP(hipStreamWaitEvent(fin, S.msm_done[k], 0));
    }
    if (xout) {
        // nothing to assemble: the ticket completes when the scalars are written
    } else if (d_partial_out) {
        ZK_HIP(hipMemsetAsync(d_partial_out, 0, ZK_PARTIAL_BYTES, fin));
        ZK_HIP(hipMemcpyAsync(d_partial_out, ms, sizeof(MsmResults), hipMemcpyDeviceToDevice, fin));
    } else {
        ZK_HIP(hipStreamWaitEvent(fin, S.pre_evt, 0));
        {
            ProfScope pscope(ctx, "assemble", 0, fin);
            hipLaunchKernelGGL(k_assemble, dim3(1), dim3(192), 0, fin, ms, &S.as.p->pre, ps.draw_blind(), S.d_proof.p);
        }
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipMemcpyAsync(S.h_proof, S.d_proof.p, ZK_PROOF_BYTES, hipMemcpyDeviceToHost, fin));
    }
    ZK_HIP(hipMemcpyAsync(S.h_flag, S.flag.p, sizeof(int), hipMemcpyDeviceToHost, fin));
    ZK_HIP(hipEventRecord(S.done_evt, fin));
    S.fin_stream = fin;
    S.busy = true;
    ctx->cur_slot = -1;
    pick_next_slot(ps);
    return ticket;
}

// The inner products of `sets` proofs over this rank's points: d_l / d_vc / d_uc / d_hb hold `sets` chunks each (what
// the all-to-all delivered: chunk j = the scalars of proof j for the points of `rank`); the partial sums of proof j go
// to d_partials_out + j ZK_PARTIAL_BYTES.  One ticket for the whole batch: the sets follow each other on the same
// five MSM streams, so one slot's workspaces serve them all.
int prove_msm_submit(zk_ctx* ctx, const zk_crs& crs_c, const zk_qap& q, int sets, int rank, int world,
                     const Fr* d_l, const Fr* d_vc, const Fr* d_uc, const Fr* d_hb, void* d_partials_out) {
    zk_crs& crs = const_cast<zk_crs&>(crs_c);
    ZK_REQUIRE(crs.n == q.n && crs.m == q.m && crs.input == q.input, ZK_ERR_ARG, "prove: CRS and QAP dimensions differ");
    ZK_REQUIRE(!q.dense, ZK_ERR_UNSUPPORTED, "prove: the scalar exchange needs a sparse QAP form");
    ProveState& ps = prove_state(ctx);
    const int ticket = ps.next;
    ProveSlot& S = ps.slot[ticket];
    ZK_REQUIRE(!S.busy, ZK_ERR_ARG, "prove: too many proofs in flight (call zk_prove_wait first)");
    const ExchangeDims xd = exchange_dims(q, world);
    // world > 1: tables of this rank's point ranges only (option rank_tables; 0 = slices of the full tables, as a lone prover has them)
    const bool rt = world > 1 && ctx->opt_rank_tables;
    // the bases follow the form the scalars were made in (prove_form: the same decision as in the scalars stage): bit-reversed powers
    // (0), Lagrange-basis points (1), the reference's own powers in natural order (2)
    const int form = prove_form(ctx, crs, q);
    if (rt) crs_ensure_rank_tables(ctx, crs, form == 0, q.log_n, form == 1, rank, world, xd.cl, xd.cn, xd.ch);
    else if (form == 1) crs_ensure_tables(ctx, crs, false, 0, true);
    else if (form == 2) crs_ensure_tables(ctx, crs, false, 0);
    else crs_ensure_tables(ctx, crs, true, q.log_n);
    StreamSwap swap_guard(ctx, (ticket & 1) ? ctx->main_alt : ctx->stream);
    hipStream_t st = ctx->stream;
    ctx->cur_slot = ticket;
    S.batch = 0;
    S.partial = true;
    const size_t n = q.n, nl = q.m > q.input + 1 ? q.m - q.input - 1 : 0, g = (size_t)rank;
    auto range = [&](size_t c, size_t count, size_t* lo) {   // this rank's points of a product: [lo, lo + returned count)
        *lo = std::min(g * c, count);
        return std::min(c, count - *lo);
    };
    if (ctx->submit_wait_evt) { ZK_HIP(hipStreamWaitEvent(st, ctx->submit_wait_evt, 0)); ctx->submit_wait_evt = nullptr; }
    ZK_HIP(hipMemsetAsync(S.flag.p, 0, sizeof(int), st));
    ZK_HIP(hipMemsetAsync(d_partials_out, 0, (size_t)sets * ZK_PARTIAL_BYTES, st));
    // one grouped product per base set: the `sets` proofs of the round share the sort, the accumulation launch and the
    // reduction tails (group j = proof j with its own 2^(c-1) buckets); chain L -> B2 -> A -> H as in a whole proof
    // base_off / rt_off: first point of the product in the whole table / in the rank's table.  scalars2 (chunk2, count2): the second
    // scalar array of the merged product L + H (MsmSplit, grouped): its points follow the first part's `chunk` points in the rank's table.
    auto launch = [&](int k, int after, auto& table, const Fr* scalars, size_t chunk, size_t count, auto* out, size_t base_off = 0, size_t rt_off = 0,
                      const Fr* scalars2 = nullptr, size_t chunk2 = 0, size_t count2 = 0) {
        hipStream_t ms_st = ctx->opt_serialize ? st : msm_stream_for(ctx, k);
        ZK_HIP(hipEventRecord(S.fork_evt, st));
        ZK_HIP(hipStreamWaitEvent(ms_st, S.fork_evt, 0));
        hipEvent_t wait_evt = after >= 0 ? S.acc_evt[after] : ps.last_acc;
        size_t lo;
        const size_t valid = range(chunk, count, &lo);
        MsmGroups grp;
        MsmSplit sp;
        size_t glen = chunk, total = valid;
        if (scalars2) {
            size_t lo2;
            const size_t valid2 = range(chunk2, count2, &lo2);
            sp.scalars2 = scalars2; sp.split = chunk; sp.n2 = valid2; sp.valid1 = valid; sp.stride1 = chunk; sp.stride2 = chunk2;
            glen = chunk + chunk2; total = chunk + valid2;
        }
        grp.groups = sets; grp.glen = glen; grp.valid = total; grp.out_stride = ZK_PARTIAL_BYTES;
        lo = rt ? rt_off : lo + base_off;   // the rank's table starts at its first point
        hipStream_t end_st = sets == 1 ? msm_run(ctx, S.ws[k], ms_st, table, scalars, total, 0, 1, out, wait_evt, S.acc_evt[k], lo, MsmGroups(), sp)
                                       : msm_run(ctx, S.ws[k], ms_st, table, scalars, 0, 0, 1, out, wait_evt, S.acc_evt[k], lo, grp, sp);
        ZK_HIP(hipEventRecord(S.msm_done[k], end_st));
        ps.last_acc = S.acc_evt[k];
    };
    if (sets > 0) {
        MsmResults* ms = reinterpret_cast<MsmResults*>(d_partials_out);
        const size_t n_hb = q.roots ? 2 * n - 1 : 2 * n;   // integer roots: L^S t/delta (n-1) | L (n)
        // With per-rank tables L joins the H product as in a whole proof (merge_lh, prove_submit): the rank's table holds its points of
        // xi_t | xi and of sum_delta back to back, the scalars come from the two exchanged arrays; ms->l stays infinity (cleared above).
        const bool merge_lh = ctx->opt_merge_lh && rt;
        if (!merge_lh) launch(1, -1, rt ? crs.rank_tabs.t_hb1 : crs.t_hb1, d_l, xd.cl, nl, &ms->l, crs.off_l, crs.rank_tabs.off_l);   // sum_delta sits behind xi_t | xi
        launch(2, merge_lh ? -1 : 1, rt ? crs.rank_tabs.t_xi1 : crs.t_xi1, d_uc, xd.cn, n, &ms->a);
        launch(0, 2, rt ? crs.rank_tabs.t_xi2 : crs.t_xi2, d_vc, xd.cn, n, &ms->b2);
        if (merge_lh) launch(4, 0, crs.rank_tabs.t_hb1, d_hb, xd.ch, n_hb, &ms->hb, 0, 0, d_l, xd.cl, nl);
        else launch(4, 0, rt ? crs.rank_tabs.t_hb1 : crs.t_hb1, d_hb, xd.ch, n_hb, &ms->hb);
    }
    hipStream_t fin = ctx->finish;
    ZK_HIP(hipEventRecord(S.fork_evt, st));
    ZK_HIP(hipStreamWaitEvent(fin, S.fork_evt, 0));
    if (sets > 0)
        for (int k = 0; k < zk_ctx::MSM_STREAMS; ++k)
            if (k != 3) ZK_HIP(hipStreamWaitEvent(fin, S.msm_done[k], 0));
    ZK_HIP(hipMemcpyAsync(S.h_flag, S.flag.p, sizeof(int), hipMemcpyDeviceToHost, fin));
    ZK_HIP(hipEventRecord(S.done_evt, fin));
    S.fin_stream = fin;
    S.busy = true;
    ctx->cur_slot = -1;
    pick_next_slot(ps);
    return ticket;
}

// `count` whole proofs over the same CRS / QAP as one batch: the SpMV / NTT stages follow each other on the main
// stream (one slot's scratch serves them all), the inner products run as ONE grouped MSM per product (proof j = group j)
// and two launches assemble all proofs.  For circuits of 2^16 gates and fewer a proof is bound by the latency of its
// ~100 dependent launches, not by arithmetic: batching spreads that chain over `count` proofs.
int prove_batch_submit(zk_ctx* ctx, const zk_crs& crs_c, const zk_qap& q, int count, const void* const* d_weights, const size_t* m_in,
                       const uint64_t* r, const uint64_t* s) {
    zk_crs& crs = const_cast<zk_crs&>(crs_c);
    ZK_REQUIRE(crs.n == q.n && crs.m == q.m && crs.input == q.input, ZK_ERR_ARG, "prove: CRS and QAP dimensions differ");
    ZK_REQUIRE(!q.dense, ZK_ERR_UNSUPPORTED, "prove: batches take the sparse QAP forms");
    ZK_REQUIRE(count >= 1 && count <= ZK_MAX_BATCH, ZK_ERR_ARG, "prove: batch size out of range");
    ProveState& ps = prove_state(ctx);
    const int ticket = ps.next;
    ProveSlot& S = ps.slot[ticket];
    ZK_REQUIRE(!S.busy, ZK_ERR_ARG, "prove: too many proofs in flight (call zk_prove_wait first)");
    const int form = prove_form(ctx, crs, q);
    if (form == 1) crs_ensure_tables(ctx, crs, false, 0, true);
    else if (form == 2) crs_ensure_tables(ctx, crs, false, 0);
    else crs_ensure_tables(ctx, crs, true, q.log_n);
    crs_ensure_fixed_tables(ctx, crs);
    if (!S.h_b_proofs) {
        ZK_HIP(hipHostMalloc((void**)&S.h_b_proofs, (size_t)ZK_MAX_BATCH * ZK_PROOF_BYTES));
        ZK_HIP(hipHostMalloc((void**)&S.h_b_rs, (size_t)ZK_MAX_BATCH * 2 * sizeof(Fr)));
    }
    for (int j = 0; j < count; ++j) {
        Fr rc = fr_from_words64(r + 4 * j), sc = fr_from_words64(s + 4 * j);
        ZK_REQUIRE(rc.raw_in_range() && sc.raw_in_range(), ZK_ERR_RANGE, "prove: r or s >= modulus");
        ZK_REQUIRE(d_weights[j], ZK_ERR_ARG, "prove: null witness");
        S.h_b_rs[2 * j] = rc; S.h_b_rs[2 * j + 1] = sc;
    }
    StreamSwap swap_guard(ctx, (ticket & 1) ? ctx->main_alt : ctx->stream);
    hipStream_t st = ctx->stream;
    ctx->cur_slot = ticket;
    S.partial = false;
    const size_t n = q.n, m = q.m, l = q.input, nl = m > l + 1 ? m - l - 1 : 0, cl = std::max<size_t>(nl, 1);
    // merge_lh (see prove_submit): the witness scalars of proof j sit behind its h | r v + s u scalars in ONE array per proof -- stride
    // hs = off_l + cl, witness at off_l -- and L joins the H product (group j = proof j); ms->l stays infinity (the blobs are cleared)
    const bool merge_lh = ctx->opt_merge_lh != 0;
    const size_t off_l = crs.off_l, hs = merge_lh ? off_l + cl : 2 * n;
    S.bx_l.ensure(merge_lh ? 1 : cl * count); S.bx_v.ensure(n * count); S.bx_u.ensure(n * count); S.bx_h.ensure(hs * count);
    S.b_rs.ensure(2 * ZK_MAX_BATCH); S.b_pre.ensure(ZK_MAX_BATCH);
    S.b_partials.ensure((size_t)ZK_MAX_BATCH * ZK_PARTIAL_BYTES); S.b_proofs.ensure((size_t)ZK_MAX_BATCH * ZK_PROOF_BYTES);
    ZK_HIP(hipMemsetAsync(S.flag.p, 0, sizeof(int), st));
    ZK_HIP(hipMemsetAsync(S.b_partials.p, 0, (size_t)count * ZK_PARTIAL_BYTES, st));
    // r/s-only fixed-base multiplications of all proofs on the side stream
    {
        hipStream_t pre_st = ctx->opt_serialize ? st : ctx->side;
        ZK_HIP(hipMemcpyAsync(S.b_rs.p, S.h_b_rs, (size_t)count * 2 * sizeof(Fr), hipMemcpyHostToDevice, pre_st));
        hipLaunchKernelGGL(k_assemble_pre_batch, dim3(count), dim3(320), 0, pre_st, crs.ft_alpha1.p, crs.ft_beta1.p, crs.ft_delta1.p, crs.ft_delta2.p,
                           crs.alpha1.p, crs.beta2.p, S.b_rs.p, S.b_pre.p);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipEventRecord(S.pre_evt, pre_st));
    }
    MsmResults* ms = reinterpret_cast<MsmResults*>(S.b_partials.p);
    auto launch = [&](int k, int after, auto& table, const Fr* scalars, size_t glen, size_t valid, auto* out, size_t off = 0) {
        hipStream_t ms_st = ctx->opt_serialize ? st : msm_stream_for(ctx, k);
        ZK_HIP(hipEventRecord(S.fork_evt, st));
        ZK_HIP(hipStreamWaitEvent(ms_st, S.fork_evt, 0));
        hipEvent_t wait_evt = after >= 0 ? S.acc_evt[after] : ps.last_acc;
        MsmGroups grp;
        grp.groups = count; grp.glen = glen; grp.valid = valid; grp.out_stride = ZK_PARTIAL_BYTES;
        hipStream_t end_st = count == 1 ? msm_run(ctx, S.ws[k], ms_st, table, scalars, valid, 0, 1, out, wait_evt, S.acc_evt[k], off)
                                        : msm_run(ctx, S.ws[k], ms_st, table, scalars, 0, 0, 1, out, wait_evt, S.acc_evt[k], off, grp);
        ZK_HIP(hipEventRecord(S.msm_done[k], end_st));
        ps.last_acc = S.acc_evt[k];
    };
    // L needs only the witnesses (zip truncation, mod.rs:233-253: zero scalars behind a short witness)
    if (merge_lh) ZK_HIP(hipMemsetAsync(S.bx_h.p, 0, hs * count * sizeof(Fr), st));
    else ZK_HIP(hipMemsetAsync(S.bx_l.p, 0, cl * count * sizeof(Fr), st));
    std::vector<size_t> a_len(count), n_l(count);
    for (int j = 0; j < count; ++j) {
        a_len[j] = std::min(m_in[j], m);
        n_l[j] = a_len[j] > l + 1 ? std::min(a_len[j] - l - 1, m - l - 1) : 0;
        Fr* dst = merge_lh ? S.bx_h.p + (size_t)j * hs + off_l : S.bx_l.p + (size_t)j * cl;
        if (n_l[j]) ZK_HIP(hipMemcpyAsync(dst, (const Fr*)d_weights[j] + l + 1, n_l[j] * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    }
    if (!merge_lh) launch(1, -1, crs.t_hb1, S.bx_l.p, cl, nl, &ms->l, crs.off_l);   // sum_delta sits behind xi_t | xi in the table
    const size_t hb_extra = merge_lh ? nl : 0;   // valid scalars of a group beyond the h | r v + s u part
    if (form == 2) {
        // arbitrary roots (arbroots.hip): every proof interpolates its own U, V (the tree's transforms are per proof); the inner products
        // of the batch run grouped like those of the other forms
        const size_t cnt = (size_t)count, amax = std::max<size_t>(*std::max_element(a_len.begin(), a_len.end()), 1);
        (void)amax;
        if (!merge_lh) ZK_HIP(hipMemsetAsync(S.bx_h.p, 0, 2 * n * cnt * sizeof(Fr), st));
        for (size_t j = 0; j < cnt; ++j) {
            fr_check_range(ctx, (const Fr*)d_weights[j], a_len[j], S.flag.p);
            arb_scalar_stage(ctx, S, q, (const Fr*)d_weights[j], a_len[j], Fr::from_canonical(S.h_b_rs[2 * j]), Fr::from_canonical(S.h_b_rs[2 * j + 1]),
                             S.bx_v.p + j * n, S.bx_u.p + j * n, S.bx_h.p + j * hs, [] {});
        }
        launch(2, 1, crs.t_xi1, S.bx_u.p, n, n, &ms->a);
        launch(0, 2, crs.t_xi2, S.bx_v.p, n, n, &ms->b2);
        launch(4, 0, crs.t_hb1, S.bx_h.p, hs, 2 * n - 1 + hb_extra, &ms->hb);      // bases: xi_t (n-1) | xi (n) [| sum_delta]; group stride hs
    } else if (form == 1) {
        // integer roots (aproots.hip): evaluation values as scalars of A and B, h on {n+1..2n-1} by one batched convolution
        const size_t cnt = (size_t)count, amax = std::max<size_t>(*std::max_element(a_len.begin(), a_len.end()), 1), M = (size_t)1 << q.ap->log_m;
        S.uv.ensure(2 * n * cnt); S.xy.ensure(3 * M * cnt);
        (void)amax;
        Fr *ve = S.uv.p, *ue = S.uv.p + n * cnt;
        for (size_t j = 0; j < cnt; ++j) {
            fr_check_range(ctx, (const Fr*)d_weights[j], a_len[j], S.flag.p);
            spmv(ctx, q.u_gate, (const Fr*)d_weights[j], a_len[j], ue + j * n);
            spmv(ctx, q.v_gate, (const Fr*)d_weights[j], a_len[j], ve + j * n);
        }
        fr_from_mont(ctx, ve, S.bx_v.p, n * cnt);
        launch(0, 1, crs.t_xi2, S.bx_v.p, n, n, &ms->b2);
        fr_from_mont(ctx, ue, S.bx_u.p, n * cnt);
        launch(2, 0, crs.t_xi1, S.bx_u.p, n, n, &ms->a);
        for (size_t j = 0; j < cnt; ++j)       // bases: L^S t/delta (n-1) | L (n); group stride 2n, 2n-1 valid
            fr_lincomb_to_canonical(ctx, ve + j * n, Fr::from_canonical(S.h_b_rs[2 * j]), ue + j * n, Fr::from_canonical(S.h_b_rs[2 * j + 1]),
                                    S.bx_h.p + j * hs + (n - 1), n);
        ap_quotient_values(ctx, q, ue, ve, S.xy.p, S.bx_h.p, cnt, hs);
        launch(4, 2, crs.t_hb1, S.bx_h.p, hs, 2 * n - 1 + hb_extra, &ms->hb);
    } else {
    auto tabs = ntt_get_tables(ctx, q.log_n);
    ntt_ensure_coset_tables(ctx, *tabs);
    // the stage of sparse_scalar_stage with a batch dimension: scratch vectors are [count][n], the element-wise kernels and
    // the six transforms cover all proofs in one launch each; what depends on a proof's own witness or (r, s) (conversion,
    // SpMV, r v + s u, h) is launched per proof: 5 count + 13 launches instead of 17 count
    const size_t cnt = (size_t)count, amax = std::max<size_t>(*std::max_element(a_len.begin(), a_len.end()), 1);
    S.uv.ensure(2 * n * cnt); S.uvg.ensure(2 * n * cnt); S.xy.ensure(2 * n * cnt);
    (void)amax;
    Fr *ve = S.uv.p, *ue = S.uv.p + n * cnt, *x0 = S.xy.p, *y0 = S.xy.p + n * cnt, *vg = S.uvg.p, *ug = S.uvg.p + n * cnt;
    for (size_t j = 0; j < cnt; ++j) {
        fr_check_range(ctx, (const Fr*)d_weights[j], a_len[j], S.flag.p);
        spmv(ctx, q.u_gate, (const Fr*)d_weights[j], a_len[j], ue + j * n);
        spmv(ctx, q.v_gate, (const Fr*)d_weights[j], a_len[j], ve + j * n);
    }
    fr_pointwise_mul(ctx, ue, ve, x0, n * cnt);                             // U.V on <w>
    const Fr n_inv = tabs->n_inv;                                           // the inverse transforms run without their 1 / n (see sparse_scalar_stage)
    ntt_dif(ctx, S.uv.p, q.log_n, true, false, 2 * cnt);                    // n V, n U coefficients (bit-reversed order)
    fr_scale_to_canonical(ctx, ve, n_inv, S.bx_v.p, n * cnt);
    launch(0, 1, crs.t_xi2, S.bx_v.p, n, n, &ms->b2);
    fr_scale_to_canonical(ctx, ue, n_inv, S.bx_u.p, n * cnt);
    launch(2, 0, crs.t_xi1, S.bx_u.p, n, n, &ms->a);
    for (size_t j = 0; j < cnt; ++j)
        fr_lincomb_to_canonical(ctx, ve + j * n, Fr::from_canonical(S.h_b_rs[2 * j]) * n_inv, ue + j * n, Fr::from_canonical(S.h_b_rs[2 * j + 1]) * n_inv,
                                S.bx_h.p + j * hs + n, n);
    ZK_HIP(hipMemcpyAsync(S.uvg.p, S.uv.p, 2 * n * cnt * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ntt_dit(ctx, S.uvg.p, q.log_n, false, false, tabs->coset_fwd_brev.p, 2 * cnt);   // V, U on g<w>
    fr_pointwise_mul(ctx, ug, vg, y0, n * cnt);                             // U.V on g<w>
    ntt_dif(ctx, S.xy.p, q.log_n, true, false, 2 * cnt);                    // n (lo + hi) | n (lo - hi)_i * g^i
    const Fr half = host_fr_from_u64(2).inv() * n_inv;
    for (size_t j = 0; j < cnt; ++j)
        h_combine(ctx, x0 + j * n, y0 + j * n, tabs->coset_inv_brev_half.p, half, S.bx_h.p + j * hs, n);
    launch(4, 2, crs.t_hb1, S.bx_h.p, hs, 2 * n + hb_extra, &ms->hb);
    }

    hipStream_t fin = ctx->finish;
    ZK_HIP(hipEventRecord(S.fork_evt, st));
    ZK_HIP(hipStreamWaitEvent(fin, S.fork_evt, 0));
    for (int k = 0; k < zk_ctx::MSM_STREAMS; ++k)
        if (k != 3) ZK_HIP(hipStreamWaitEvent(fin, S.msm_done[k], 0));
    ZK_HIP(hipStreamWaitEvent(fin, S.pre_evt, 0));
    {
        ProfScope pscope(ctx, "assemble", 0, fin);
        hipLaunchKernelGGL(k_assemble_batch, dim3(count), dim3(192), 0, fin, S.b_partials.p, S.b_pre.p, ps.draw_blind(), S.b_proofs.p);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(S.h_b_proofs, S.b_proofs.p, (size_t)count * ZK_PROOF_BYTES, hipMemcpyDeviceToHost, fin));
    ZK_HIP(hipMemcpyAsync(S.h_flag, S.flag.p, sizeof(int), hipMemcpyDeviceToHost, fin));
    ZK_HIP(hipEventRecord(S.done_evt, fin));
    S.fin_stream = fin;
    S.batch = count;
    S.busy = true;
    ctx->cur_slot = -1;
    pick_next_slot(ps);
    return ticket;
}

void prove_batch_wait(zk_ctx* ctx, int ticket, int count, uint8_t* proofs_out) {
    ProveState& ps = prove_state(ctx);
    ZK_REQUIRE(ticket >= 0 && ticket < ProveState::SLOTS && ps.slot[ticket].busy, ZK_ERR_ARG, "prove_wait: no proof in flight for this ticket");
    ProveSlot& S = ps.slot[ticket];
    ZK_REQUIRE(S.batch == count, ZK_ERR_ARG, "prove_batch_wait: the ticket belongs to a batch of a different size");
    S.busy = false;
    S.batch = 0;
    ZK_HIP(hipEventSynchronize(S.done_evt));
    pick_next_slot(ps);
    ctx->resolve_profile(ticket);
    ZK_REQUIRE(!*S.h_flag, ZK_ERR_RANGE, "prove: witness element >= r");
    std::memcpy(proofs_out, S.h_b_proofs, (size_t)count * ZK_PROOF_BYTES);
}

// Waits for a submitted proof; proof_out may be null for partial submissions.
void prove_wait(zk_ctx* ctx, int ticket, uint8_t* proof_out) {
    ProveState& ps = prove_state(ctx);
    ZK_REQUIRE(ticket >= 0 && ticket < ProveState::SLOTS && ps.slot[ticket].busy, ZK_ERR_ARG, "prove_wait: no proof in flight for this ticket");
    ProveSlot& S = ps.slot[ticket];
    ZK_REQUIRE(S.batch == 0, ZK_ERR_ARG, "prove_wait: batch ticket (use zk_prove_batch_wait)");
    S.busy = false;
    ZK_HIP(hipEventSynchronize(S.done_evt));
    pick_next_slot(ps);
    ctx->resolve_profile(ticket);
    ZK_REQUIRE(!*S.h_flag, ZK_ERR_RANGE, "prove: witness element >= r");
    if (!S.partial && proof_out) std::memcpy(proof_out, S.h_proof, ZK_PROOF_BYTES);
}

hipStream_t prove_ticket_stream(zk_ctx* ctx, int ticket) {
    ProveState& ps = prove_state(ctx);
    ZK_REQUIRE(ticket >= 0 && ticket < ProveState::SLOTS && ps.slot[ticket].busy, ZK_ERR_ARG, "prove: no proof in flight for this ticket");
    return ps.slot[ticket].fin_stream;
}

// Frees the slot of a ticket whose work is enqueued but not necessarily complete.  Safe because of the stream discipline of this
// file: whatever a later ticket on the same slot enqueues goes onto the same main stream (slot parity), the same inner-product
// streams and the same finish stream as this ticket's work, behind it.  The caller orders its own consumers with an event it records
// on prove_ticket_stream().
void prove_release(zk_ctx* ctx, int ticket, int* h_flag_pinned) {
    ProveState& ps = prove_state(ctx);
    ZK_REQUIRE(ticket >= 0 && ticket < ProveState::SLOTS && ps.slot[ticket].busy, ZK_ERR_ARG, "prove: no proof in flight for this ticket");
    ProveSlot& S = ps.slot[ticket];
    ZK_REQUIRE(S.batch == 0, ZK_ERR_ARG, "prove_release: batch ticket");
    if (h_flag_pinned) ZK_HIP(hipMemcpyAsync(h_flag_pinned, S.flag.p, sizeof(int), hipMemcpyDeviceToHost, S.fin_stream));
    S.busy = false;
    pick_next_slot(ps);
}

void prove_dev(zk_ctx* ctx, const zk_crs& crs, const zk_qap& qap, const Fr* d_weights, size_t m, const uint64_t* r, const uint64_t* s,
               uint8_t* proof_out, int rank, int world, void* d_partial_out) {
    int t = prove_submit(ctx, crs, qap, d_weights, m, r, s, rank, world, d_partial_out, nullptr);
    prove_wait(ctx, t, proof_out);
}

// The witness is in HOST memory: it is copied into the slot's own device buffer on the stream the proof's first
// kernels run on, so with page-locked memory (zk_host_alloc) the transfer of proof k+1 overlaps the inner products of
// proof k and the PCIe-inclusive rate of a stream of proofs is the resident rate.  Pageable memory works too (the
// runtime stages it; the call then blocks for the duration of the copy).
int prove_submit_host(zk_ctx* ctx, const zk_crs& crs, const zk_qap& qap, const uint64_t* weights, size_t m, const uint64_t* r, const uint64_t* s,
                      int world, Fr* const* xout) {
    ProveState& ps = prove_state(ctx);
    ProveSlot& S = ps.slot[ps.next];
    ZK_REQUIRE(!S.busy, ZK_ERR_ARG, "prove: too many proofs in flight (call zk_prove_wait first)");
    const size_t mm = std::min(m, qap.m);   // zip truncation (mod.rs:233-253): elements behind m_qap are never read
    S.d_wit.ensure(std::max<size_t>(mm, 1));
    hipStream_t st = (ps.next & 1) ? ctx->main_alt : ctx->stream;
    if (mm) ZK_HIP(hipMemcpyAsync(S.d_wit.p, weights, mm * sizeof(Fr), hipMemcpyHostToDevice, st));
    return prove_submit(ctx, crs, qap, S.d_wit.p, mm, r, s, 0, world, nullptr, xout);   // xout: scalars only (multi-GPU exchange)
}

void prove_host(zk_ctx* ctx, const zk_crs& crs, const zk_qap& qap, const uint64_t* weights, size_t m, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out) {
    int t = prove_submit_host(ctx, crs, qap, weights, m, r, s);
    prove_wait(ctx, t, proof_out);
}

void prove_combine_on(zk_ctx* ctx, const zk_crs& crs_c, const void* d_partials, int world, const uint64_t r[4], const uint64_t s[4], hipStream_t st,
                      uint8_t* h_proof_pinned) {
    zk_crs& crs = const_cast<zk_crs&>(crs_c);
    Fr rc = fr_from_words64(r), sc = fr_from_words64(s);
    ZK_REQUIRE(rc.raw_in_range() && sc.raw_in_range(), ZK_ERR_RANGE, "prove: r or s >= modulus");
    crs_ensure_fixed_tables(ctx, crs);
    ProveState& ps = prove_state(ctx);
    launch_pre(ctx, crs, st, rc, sc, ps.comb_as.p);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(320), 0, st, (const uint8_t*)d_partials, world, ps.comb_ms.p);
    hipLaunchKernelGGL(k_assemble, dim3(1), dim3(192), 0, st, ps.comb_ms.p, &ps.comb_as.p->pre, ps.draw_blind(), ps.comb_proof.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(h_proof_pinned, ps.comb_proof.p, ZK_PROOF_BYTES, hipMemcpyDeviceToHost, st));
}

void prove_combine(zk_ctx* ctx, const zk_crs& crs, const void* d_partials, int world, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out) {
    ProveState& ps = prove_state(ctx);
    // on the side stream: a pipelined caller has the next proofs' stages queued on the main streams already, and the
    // finish stream holds their joins (it would make this proof's assembly wait for the NEXT proof's inner products)
    hipStream_t st = ctx->side;
    prove_combine_on(ctx, crs, d_partials, world, r, s, st, ps.comb_h_proof);
    ZK_HIP(hipStreamSynchronize(st));
    std::memcpy(proof_out, ps.comb_h_proof, ZK_PROOF_BYTES);
}

}  // namespace zk
