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
#include "pipeline.hpp"
#include "qap_kernels.hpp"

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
    G1J r_delta;      // r * delta1
    G1J fixed_c;      // s * alpha1 + r * beta1 + (r s) * delta1
    G2J s_delta2;     // s * delta2
};

__device__ __forceinline__ uint32_t nibble(const Fr& k, int w) { return (k.l[w >> 3] >> ((w & 7) * 4)) & 15u; }

// k * P from the 4-bit fixed-base table FT[w][d] = d * 16^w * P: lane w contributes FT[w][digit_w];
// the 64 contributions are summed by a tree over LDS (6 additions deep instead of 254 doublings).
template <class F>
__device__ void fixed_base_mul_wave(const Aff<F>* __restrict__ ft, const Fr& k, Jac<F>* sh, int lane, Jac<F>* out) {
    sh[lane] = Jac<F>::from_affine(ft[lane * 16 + nibble(k, lane)]);
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) {
        if (lane < d) sh[lane] = jac_add_ni(sh[lane], sh[lane + d]);
        __syncthreads();
    }
    if (lane == 0) *out = sh[0];
}

// Everything that depends only on (r, s) and single CRS points; runs on the side stream while the
// five inner products execute.  One wave per fixed-base multiplication.
__global__ __launch_bounds__(320) void k_assemble_pre(const G1A* __restrict__ ft_alpha1, const G1A* __restrict__ ft_beta1,
                                                      const G1A* __restrict__ ft_delta1, const G2A* __restrict__ ft_delta2,
                                                      Fr r, Fr s, AssemblePre* __restrict__ out) {
    __shared__ G1J sh1[4][64];
    __shared__ G2J sh2[64];
    __shared__ G1J res[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Fr rs = (Fr::from_canonical(r) * Fr::from_canonical(s)).to_canonical();
    // every wave reaches the same number of __syncthreads (7) inside fixed_base_mul_wave
    if (wave == 0) fixed_base_mul_wave<Fq>(ft_delta1, r, sh1[0], lane, &res[0]);          // r delta
    else if (wave == 1) fixed_base_mul_wave<Fq>(ft_alpha1, s, sh1[1], lane, &res[1]);     // s alpha
    else if (wave == 2) fixed_base_mul_wave<Fq>(ft_beta1, r, sh1[2], lane, &res[2]);      // r beta
    else if (wave == 3) fixed_base_mul_wave<Fq>(ft_delta1, rs, sh1[3], lane, &res[3]);    // rs delta
    else fixed_base_mul_wave<Fq2>(ft_delta2, s, sh2, lane, &out->s_delta2);               // s delta2
    __syncthreads();
    if (threadIdx.x == 0) {
        out->r_delta = res[0];
        out->fixed_c = jac_add_ni(jac_add_ni(res[1], res[2]), res[3]);
    }
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
__device__ void encode_g1(const G1J& p, uint8_t* out) {
    for (int i = 0; i < 65; ++i) out[i] = 0;
    if (p.is_inf()) return;
    G1A a = jac_to_affine(p);
    out[0] = 4;
    put_be32(a.x, out + 1);
    put_be32(a.y, out + 33);
}
__device__ void encode_g2(const G2J& p, uint8_t* out) {
    for (int i = 0; i < 129; ++i) out[i] = 0;
    if (p.is_inf()) return;
    G2A a = jac_to_affine(p);
    out[0] = 4;
    put_be32(a.x.c1, out + 1);
    put_be32(a.x.c0, out + 33);
    put_be32(a.y.c1, out + 65);
    put_be32(a.y.c0, out + 97);
}

// (mod.rs:274-293)  a = A + alpha + r delta ;  b = B2 + beta2 + s delta2 ;
// c = H + L + s a + r (beta + B1 + s delta) - (r s) delta
//   = [H + r B1 + s A] + L + [s alpha + r beta + (r s) delta]
// where H + r B1 + s A comes out of ONE inner product: scalars h_i over xi_t and (r v_i + s u_i) over
// xi.  No scalar multiplication with a run-time base is left.
__global__ __launch_bounds__(192) void k_assemble(const MsmResults* __restrict__ ms, const AssemblePre* __restrict__ pre,
                                                  const G1A* __restrict__ alpha1, const G2A* __restrict__ beta2, uint8_t* __restrict__ proof) {
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x & 63) return;
    if (wave == 0) encode_g1(jac_add_ni(jac_madd_ni(ms->a, *alpha1), pre->r_delta), proof);
    if (wave == 1) encode_g2(jac_add_ni(jac_madd_ni(ms->b2, *beta2), pre->s_delta2), proof + 65);
    if (wave == 2) encode_g1(jac_add_ni(jac_add_ni(ms->hb, ms->l), pre->fixed_c), proof + 65 + 129);
}

__global__ void k_sum_partials(const uint8_t* __restrict__ partials, int world, MsmResults* __restrict__ out) {
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

static void launch_pre(zk_ctx* ctx, const zk_crs& crs, hipStream_t st, const Fr& rc, const Fr& sc, AssembleScratch* d_as) {
    hipLaunchKernelGGL(k_assemble_pre, dim3(1), dim3(320), 0, st, crs.ft_alpha1.p, crs.ft_beta1.p, crs.ft_delta1.p, crs.ft_delta2.p, rc, sc, &d_as->pre);
    ZK_HIP(hipGetLastError());
}
// final additions + affine normalisation + canonical encoding, then copy the 259 bytes out
static void finish(zk_ctx* ctx, const zk_crs& crs, const MsmResults* d_ms, AssembleScratch* d_as, uint8_t* d_proof, uint8_t* proof_out) {
    hipStream_t st = ctx->stream;
    {
        ProfScope ps(ctx, "assemble", 0);
        hipLaunchKernelGGL(k_assemble, dim3(1), dim3(192), 0, st, d_ms, &d_as->pre, crs.alpha1.p, crs.beta2.p, d_proof);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(proof_out, d_proof, ZK_PROOF_BYTES, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
}

void prove_dev(zk_ctx* ctx, const zk_crs& crs_c, const zk_qap& qap_c, const Fr* d_weights, size_t m_in, const uint64_t* r, const uint64_t* s,
               uint8_t* proof_out, int rank, int world, void* d_partial_out) {
    zk_crs& crs = const_cast<zk_crs&>(crs_c);   // lazily built caches / scratch only
    zk_qap& q = const_cast<zk_qap&>(qap_c);
    ZK_REQUIRE(crs.n == q.n && crs.m == q.m && crs.input == q.input, ZK_ERR_ARG, "prove: CRS and QAP dimensions differ");
    ZK_REQUIRE(d_partial_out || world == 1, ZK_ERR_ARG, "prove: world > 1 needs a partial output buffer");
    const size_t n = q.n, m = q.m, l = q.input;
    const size_t a_len = std::min(m_in, m);   // zip(weights) truncates (mod.rs:233-253)
    hipStream_t st = ctx->stream;

    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    q.a_mont.ensure(std::max<size_t>(a_len, 1));
    fr_to_mont(ctx, d_weights, q.a_mont.p, a_len, flag.p);

    // the r/s-only fixed-base multiplications overlap with everything below on the side stream
    DevBuf<AssembleScratch> d_as(1);
    DevBuf<uint8_t> d_proof(ZK_PROOF_BYTES);
    Fr rc = fr_from_words64(r), sc = fr_from_words64(s);
    ZK_REQUIRE(rc.raw_in_range() && sc.raw_in_range(), ZK_ERR_RANGE, "prove: r or s >= modulus");
    const Fr r_mont = Fr::from_canonical(rc), s_mont = Fr::from_canonical(sc);
    hipEvent_t pre_evt = nullptr;
    if (!d_partial_out) {
        crs_ensure_fixed_tables(ctx, crs);
        pre_evt = ctx->get_event();
        launch_pre(ctx, crs, ctx->side, rc, sc, d_as.p);
        ZK_HIP(hipEventRecord(pre_evt, ctx->side));
    }

    // The five inner products run on their own streams; each is forked from the main stream as
    // soon as its scalars exist (L needs only the witness, A only sum a_i u_i, ...), so the NTT stage
    // and the latency-bound reduction tails hide behind bucket accumulation.  Joined before assembly.
    DevBuf<MsmResults> d_ms(1);
    MsmResults* ms = d_ms.p;
    const size_t n_l = a_len > l + 1 ? std::min(a_len - l - 1, m - l - 1) : 0;
    // `after`: slot whose accumulation must finish first (-1: none).  The accumulation kernels are
    // chained (see the call sites): the long G2 kernel runs while the rest of the NTT stage proceeds at
    // high priority, its long reduction tail hides behind the G1 accumulations, and the last tail
    // exposed is the short G1 one.
    auto launch = [&](int k, int after, auto& table, const Fr* scalars, size_t count, auto* out) {
        hipStream_t ms_st = ctx->msm_stream[k];
        if (!ctx->msm_ws[k]) ctx->msm_ws[k] = std::make_shared<MsmWorkspace>();
        ZK_HIP(hipEventRecord(ctx->fork_evt, st));
        ZK_HIP(hipStreamWaitEvent(ms_st, ctx->fork_evt, 0));
        msm_run(ctx, *ctx->msm_ws[k], ms_st, table, scalars, count, rank, world, out,
                after >= 0 ? ctx->acc_evt[after] : (after == -2 ? ctx->ntt_done : nullptr), ctx->acc_evt[k]);
        ZK_HIP(hipEventRecord(ctx->msm_done[k], ms_st));
    };
    if (!q.dense) {
        crs_ensure_tables(ctx, crs, true, q.log_n);
        auto tabs = ntt_get_tables(ctx, q.log_n);
        ntt_ensure_coset_tables(ctx, *tabs);
        q.ue.ensure(n); q.ve.ensure(n); q.x0.ensure(n); q.y0.ensure(n); q.ug.ensure(n); q.vg.ensure(n);
        q.uc_can.ensure(n); q.vc_can.ensure(n); q.hb_can.ensure(2 * n);
        // accumulation chain L -> B2 -> A -> H+rB1+sA: L needs only the witness, so the chip is busy
        // ~0.6 ms after the call starts; the long G2 reduction tail hides behind A and the H product
        launch(1, -1, crs.t_sum_delta1, d_weights + l + 1, n_l, &ms->l);  // L: sum a_i * sum_delta_i
        spmv(ctx, q.u_gate, q.a_mont.p, a_len, q.ue.p);
        spmv(ctx, q.v_gate, q.a_mont.p, a_len, q.ve.p);
        fr_pointwise_mul(ctx, q.ue.p, q.ve.p, q.x0.p, n);                 // U.V on <w>
        ntt_dif(ctx, q.ve.p, q.log_n, true, true);                        // V coefficients (bit-reversed order)
        fr_from_mont(ctx, q.ve.p, q.vc_can.p, n);
        launch(0, 1, crs.t_xi2, q.vc_can.p, n, &ms->b2);                  // B in G2
        ntt_dif(ctx, q.ue.p, q.log_n, true, true);                        // U coefficients
        fr_from_mont(ctx, q.ue.p, q.uc_can.p, n);
        launch(2, 0, crs.t_xi1, q.uc_can.p, n, &ms->a);                   // A
        // r v_i + s u_i: B in G1 (needed only as r*B1) and s*A are folded into the H product as scalars
        fr_lincomb_to_canonical(ctx, q.ve.p, r_mont, q.ue.p, s_mont, q.hb_can.p + n, n);
        ZK_HIP(hipMemcpyAsync(q.ug.p, q.ue.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipMemcpyAsync(q.vg.p, q.ve.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ntt_dit(ctx, q.ug.p, q.log_n, false, false, tabs->coset_fwd_brev.p);   // U on g<w>
        ntt_dit(ctx, q.vg.p, q.log_n, false, false, tabs->coset_fwd_brev.p);
        fr_pointwise_mul(ctx, q.ug.p, q.vg.p, q.y0.p, n);                 // U.V on g<w>
        ntt_dif(ctx, q.x0.p, q.log_n, true, true);                        // lo + hi
        ntt_dif(ctx, q.y0.p, q.log_n, true, true);                        // (lo - hi)_i * g^i
        Fr half = host_fr_from_u64(2).inv();
        h_combine(ctx, q.x0.p, q.y0.p, tabs->coset_inv_brev_half.p, half, q.hb_can.p, n);
        // bases: xi_t (n entries, entry brev(n-1) = n-1 is infinity) | xi (n entries)
        launch(4, 2, crs.t_hb1, q.hb_can.p, 2 * n, &ms->hb);              // H + r B1 + s A: last in the chain
    } else {
        ZK_REQUIRE(!q.t_is_zero, ZK_ERR_DIV_BY_ZERO, "Dividend must be non-zero");   // field/mod.rs:440
        crs_ensure_tables(ctx, crs, false, 0);
        unsigned lc = 1;
        while (((size_t)1 << lc) < 2 * n) ++lc;
        size_t nc = (size_t)1 << lc;
        q.ue.ensure(n); q.ve.ensure(n); q.wc.ensure(n); q.prod_a.ensure(nc); q.prod_b.ensure(nc);
        q.uc_can.ensure(n); q.vc_can.ensure(n); q.hb_can.ensure(2 * n);
        launch(1, -1, crs.t_sum_delta1, d_weights + l + 1, n_l, &ms->l);
        dense_matvec(ctx, q.du.p, q.a_mont.p, a_len, n, q.ue.p);
        dense_matvec(ctx, q.dv.p, q.a_mont.p, a_len, n, q.ve.p);
        dense_matvec(ctx, q.dw.p, q.a_mont.p, a_len, n, q.wc.p);
        fr_from_mont(ctx, q.ue.p, q.uc_can.p, n);
        fr_from_mont(ctx, q.ve.p, q.vc_can.p, n);
        launch(2, -1, crs.t_xi1, q.uc_can.p, n, &ms->a);
        launch(0, -1, crs.t_xi2, q.vc_can.p, n, &ms->b2);
        fr_lincomb_to_canonical(ctx, q.ve.p, r_mont, q.ue.p, s_mont, q.hb_can.p + (n - 1), n);   // bases: xi_t (n-1) | xi (n)
        ZK_HIP(hipMemsetAsync(q.prod_a.p, 0, nc * sizeof(Fr), st));
        ZK_HIP(hipMemsetAsync(q.prod_b.p, 0, nc * sizeof(Fr), st));
        ZK_HIP(hipMemcpyAsync(q.prod_a.p, q.ue.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipMemcpyAsync(q.prod_b.p, q.ve.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ntt_dif(ctx, q.prod_a.p, lc, false, false);
        ntt_dif(ctx, q.prod_b.p, lc, false, false);
        fr_pointwise_mul(ctx, q.prod_a.p, q.prod_b.p, q.prod_a.p, nc);
        ntt_dit(ctx, q.prod_a.p, lc, true, true, nullptr);                // U*V coefficients, natural order
        fr_sub_inplace(ctx, q.prod_a.p, q.wc.p, n);                       // - W
        // quotient by t (degree d); remainder dropped (coefficient_poly.rs:148-157)
        ZK_HIP(hipMemsetAsync(q.prod_b.p, 0, nc * sizeof(Fr), st));
        size_t len_r = 2 * n - 1, d = q.t_degree;
        if (len_r > d) poly_divide(ctx, q.prod_a.p, len_r, q.dt.p, d, q.t_cinv.p, q.prod_b.p);
        fr_from_mont(ctx, q.prod_b.p, q.hb_can.p, n - 1);
        launch(4, -1, crs.t_hb1, q.hb_can.p, 2 * n - 1, &ms->hb);
    }
    for (int k = 0; k < zk_ctx::MSM_STREAMS; ++k) ZK_HIP(hipStreamWaitEvent(st, ctx->msm_done[k], 0));

    int hflag = 0;
    ZK_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    if (d_partial_out) {
        ZK_HIP(hipMemsetAsync(d_partial_out, 0, ZK_PARTIAL_BYTES, st));
        ZK_HIP(hipMemcpyAsync(d_partial_out, ms, sizeof(MsmResults), hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipStreamSynchronize(st));
        ZK_REQUIRE(!hflag, ZK_ERR_RANGE, "prove: witness element >= r");
        return;
    }
    ZK_HIP(hipStreamWaitEvent(st, pre_evt, 0));
    ctx->event_pool.push_back(pre_evt);
    finish(ctx, crs, ms, d_as.p, d_proof.p, proof_out);
    ZK_REQUIRE(!hflag, ZK_ERR_RANGE, "prove: witness element >= r");
}

void prove_host(zk_ctx* ctx, const zk_crs& crs, const zk_qap& qap, const uint64_t* weights, size_t m, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out) {
    DevBuf<Fr> dw(std::max<size_t>(m, 1));
    if (m) ZK_HIP(hipMemcpyAsync(dw.p, weights, m * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    prove_dev(ctx, crs, qap, dw.p, m, r, s, proof_out, 0, 1, nullptr);
}

void prove_combine(zk_ctx* ctx, const zk_crs& crs_c, const void* d_partials, int world, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out) {
    zk_crs& crs = const_cast<zk_crs&>(crs_c);
    Fr rc = fr_from_words64(r), sc = fr_from_words64(s);
    ZK_REQUIRE(rc.raw_in_range() && sc.raw_in_range(), ZK_ERR_RANGE, "prove: r or s >= modulus");
    crs_ensure_fixed_tables(ctx, crs);
    DevBuf<MsmResults> d_ms(1);
    DevBuf<AssembleScratch> d_as(1);
    DevBuf<uint8_t> d_proof(ZK_PROOF_BYTES);
    launch_pre(ctx, crs, ctx->stream, rc, sc, d_as.p);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(320), 0, ctx->stream, (const uint8_t*)d_partials, world, d_ms.p);
    ZK_HIP(hipGetLastError());
    finish(ctx, crs, d_ms.p, d_as.p, d_proof.p, proof_out);
}

}  // namespace zk
