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
    G1J a, b1, h, l;
    G2J b2;
};
static_assert(sizeof(MsmResults) == 4 * 96 + 192, "partial layout");
static_assert(sizeof(MsmResults) <= ZK_PARTIAL_BYTES, "ZK_PARTIAL_BYTES too small");

struct AssemblePre {
    G1J r_delta, s_delta, rs_delta;   // r*delta1, s*delta1, (r*s)*delta1
    G2J s_delta2;                     // s*delta2
};

// independent of the MSMs: runs on the side stream while they execute
__global__ __launch_bounds__(256) void k_assemble_pre(const G1A* __restrict__ delta1, const G2A* __restrict__ delta2, Fr r, Fr s, AssemblePre* __restrict__ out) {
    int wave = threadIdx.x >> 6;
    if (threadIdx.x & 63) return;
    if (wave == 0) out->r_delta = jac_mul_words(G1J::from_affine(*delta1), r.l);
    if (wave == 1) out->s_delta2 = jac_mul_words(G2J::from_affine(*delta2), s.l);
    if (wave == 2) out->s_delta = jac_mul_words(G1J::from_affine(*delta1), s.l);
    if (wave == 3) {
        Fr rs = (Fr::from_canonical(r) * Fr::from_canonical(s)).to_canonical();
        out->rs_delta = jac_mul_words(G1J::from_affine(*delta1), rs.l);
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

// a = A + alpha + r delta ; b = B2 + beta2 + s delta2 ;
// c = H + L + s a + r (beta + B1 + s delta) - (r s) delta            (mod.rs:274-293)
__global__ __launch_bounds__(256) void k_assemble(const MsmResults* __restrict__ ms, const AssemblePre* __restrict__ pre,
                                                  const G1A* __restrict__ alpha1, const G1A* __restrict__ beta1, const G2A* __restrict__ beta2,
                                                  Fr r, Fr s, uint8_t* __restrict__ proof) {
    __shared__ G1J sa, rb;
    int wave = threadIdx.x >> 6;
    bool lead = (threadIdx.x & 63) == 0;
    if (lead && wave == 0) {
        G1J a = jac_add_ni(jac_madd_ni(ms->a, *alpha1), pre->r_delta);
        encode_g1(a, proof);
        sa = jac_mul_words(a, s.l);
    }
    if (lead && wave == 1) {
        G1J t = jac_add_ni(jac_madd_ni(ms->b1, *beta1), pre->s_delta);
        rb = jac_mul_words(t, r.l);
    }
    if (lead && wave == 2) {
        G2J b = jac_add_ni(jac_madd_ni(ms->b2, *beta2), pre->s_delta2);
        encode_g2(b, proof + 65);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        G1J c = jac_add_ni(jac_add_ni(jac_add_ni(ms->h, ms->l), jac_add_ni(sa, rb)), pre->rs_delta.neg());
        encode_g1(c, proof + 65 + 129);
    }
}

__global__ void k_sum_partials(const uint8_t* __restrict__ partials, int world, MsmResults* __restrict__ out) {
    int which = threadIdx.x >> 6;
    if ((threadIdx.x & 63) || which > 4) return;
    if (which < 4) {
        G1J acc = G1J::infinity();
        for (int g = 0; g < world; ++g) {
            const MsmResults* p = reinterpret_cast<const MsmResults*>(partials + (size_t)g * ZK_PARTIAL_BYTES);
            const G1J* src = which == 0 ? &p->a : which == 1 ? &p->b1 : which == 2 ? &p->h : &p->l;
            acc = jac_add_ni(acc, *src);
        }
        G1J* dst = which == 0 ? &out->a : which == 1 ? &out->b1 : which == 2 ? &out->h : &out->l;
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

static void finish(zk_ctx* ctx, const zk_crs& crs, const MsmResults* d_ms, const uint64_t r[4], const uint64_t s[4],
                   AssemblePre* d_pre, bool pre_done, uint8_t* proof_out) {
    Fr rc = fr_from_words64(r), sc = fr_from_words64(s);
    ZK_REQUIRE(rc.raw_in_range() && sc.raw_in_range(), ZK_ERR_RANGE, "prove: r or s >= modulus");
    hipStream_t st = ctx->stream;
    if (!pre_done) {
        ProfScope ps(ctx, "assemble_pre", 0);
        hipLaunchKernelGGL(k_assemble_pre, dim3(1), dim3(256), 0, st, crs.delta1.p, crs.delta2.p, rc, sc, d_pre);
    }
    DevBuf<uint8_t> d_proof(ZK_PROOF_BYTES);
    {
        ProfScope ps(ctx, "assemble", 0);
        hipLaunchKernelGGL(k_assemble, dim3(1), dim3(256), 0, st, d_ms, d_pre, crs.alpha1.p, crs.beta1.p, crs.beta2.p, rc, sc, d_proof.p);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(proof_out, d_proof.p, ZK_PROOF_BYTES, hipMemcpyDeviceToHost, st));
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

    // the r/s-only scalar multiplications overlap with everything below on the side stream
    DevBuf<AssemblePre> d_pre(1);
    hipEvent_t pre_evt = nullptr;
    if (!d_partial_out) {
        Fr rc = fr_from_words64(r), sc = fr_from_words64(s);
        ZK_REQUIRE(rc.raw_in_range() && sc.raw_in_range(), ZK_ERR_RANGE, "prove: r or s >= modulus");
        pre_evt = ctx->get_event();
        hipLaunchKernelGGL(k_assemble_pre, dim3(1), dim3(256), 0, ctx->side, crs.delta1.p, crs.delta2.p, rc, sc, d_pre.p);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipEventRecord(pre_evt, ctx->side));
    }

    const Fr *uc_can, *vc_can, *h_can;
    size_t n_h;   // number of h coefficients paired with xi_t
    if (!q.dense) {
        crs_ensure_tables(ctx, crs, true, q.log_n);
        auto tabs = ntt_get_tables(ctx, q.log_n);
        ntt_ensure_coset_tables(ctx, *tabs);
        q.ue.ensure(n); q.ve.ensure(n); q.x0.ensure(n); q.y0.ensure(n); q.ug.ensure(n); q.vg.ensure(n);
        q.uc_can.ensure(n); q.vc_can.ensure(n); q.h_can.ensure(n);
        spmv(ctx, q.u_gate, q.a_mont.p, a_len, q.ue.p);
        spmv(ctx, q.v_gate, q.a_mont.p, a_len, q.ve.p);
        fr_pointwise_mul(ctx, q.ue.p, q.ve.p, q.x0.p, n);                 // U.V on <w>
        ntt_dif(ctx, q.ue.p, q.log_n, true, true);                        // U coefficients (bit-reversed order)
        ntt_dif(ctx, q.ve.p, q.log_n, true, true);
        ZK_HIP(hipMemcpyAsync(q.ug.p, q.ue.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipMemcpyAsync(q.vg.p, q.ve.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ntt_dit(ctx, q.ug.p, q.log_n, false, false, tabs->coset_fwd_brev.p);   // U on g<w>
        ntt_dit(ctx, q.vg.p, q.log_n, false, false, tabs->coset_fwd_brev.p);
        fr_pointwise_mul(ctx, q.ug.p, q.vg.p, q.y0.p, n);                 // U.V on g<w>
        ntt_dif(ctx, q.x0.p, q.log_n, true, true);                        // lo + hi
        ntt_dif(ctx, q.y0.p, q.log_n, true, true);                        // (lo - hi)_i * g^i
        Fr half = host_fr_from_u64(2).inv();
        h_combine(ctx, q.x0.p, q.y0.p, tabs->coset_inv_brev_half.p, half, q.h_can.p, n);
        fr_from_mont(ctx, q.ue.p, q.uc_can.p, n);
        fr_from_mont(ctx, q.ve.p, q.vc_can.p, n);
        uc_can = q.uc_can.p; vc_can = q.vc_can.p; h_can = q.h_can.p;
        n_h = n;   // entry brev(n-1) = n-1 of the bit-reversed xi_t table is infinity
    } else {
        ZK_REQUIRE(!q.t_is_zero, ZK_ERR_DIV_BY_ZERO, "Dividend must be non-zero");   // field/mod.rs:440
        crs_ensure_tables(ctx, crs, false, 0);
        unsigned lc = 1;
        while (((size_t)1 << lc) < 2 * n) ++lc;
        size_t nc = (size_t)1 << lc;
        q.ue.ensure(n); q.ve.ensure(n); q.wc.ensure(n); q.prod_a.ensure(nc); q.prod_b.ensure(nc);
        q.uc_can.ensure(n); q.vc_can.ensure(n); q.h_can.ensure(nc);
        dense_matvec(ctx, q.du.p, q.a_mont.p, a_len, n, q.ue.p);
        dense_matvec(ctx, q.dv.p, q.a_mont.p, a_len, n, q.ve.p);
        dense_matvec(ctx, q.dw.p, q.a_mont.p, a_len, n, q.wc.p);
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
        n_h = n - 1;
        fr_from_mont(ctx, q.prod_b.p, q.h_can.p, n_h);
        fr_from_mont(ctx, q.ue.p, q.uc_can.p, n);
        fr_from_mont(ctx, q.ve.p, q.vc_can.p, n);
        uc_can = q.uc_can.p; vc_can = q.vc_can.p; h_can = q.h_can.p;
    }

    // the five inner products run on their own streams: the latency-bound reduction tail of one
    // overlaps the accumulation of the others
    DevBuf<MsmResults> d_ms(1);
    MsmResults* ms = d_ms.p;
    const size_t n_l = a_len > l + 1 ? std::min(a_len - l - 1, m - l - 1) : 0;
    ZK_HIP(hipEventRecord(ctx->fork_evt, st));
    auto launch = [&](int k, auto& table, const Fr* scalars, size_t count, auto* out) {
        hipStream_t ms_st = ctx->msm_stream[k];
        if (!ctx->msm_ws[k]) ctx->msm_ws[k] = std::make_shared<MsmWorkspace>();
        ZK_HIP(hipStreamWaitEvent(ms_st, ctx->fork_evt, 0));
        msm_run(ctx, *ctx->msm_ws[k], ms_st, table, scalars, count, rank, world, out);
        ZK_HIP(hipEventRecord(ctx->msm_done[k], ms_st));
        ZK_HIP(hipStreamWaitEvent(st, ctx->msm_done[k], 0));
    };
    launch(0, crs.t_xi2, vc_can, n, &ms->b2);                       // B in G2 first: longest tail
    launch(1, crs.t_sum_delta1, d_weights + l + 1, n_l, &ms->l);    // L: sum a_i * sum_delta_i
    launch(2, crs.t_xi1, uc_can, n, &ms->a);                        // A
    launch(3, crs.t_xi1, vc_can, n, &ms->b1);                       // B in G1
    launch(4, crs.t_xi_t1, h_can, n_h, &ms->h);                     // H: sum h_i * xi_t_i

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
    finish(ctx, crs, ms, r, s, d_pre.p, true, proof_out);
    ZK_REQUIRE(!hflag, ZK_ERR_RANGE, "prove: witness element >= r");
}

void prove_host(zk_ctx* ctx, const zk_crs& crs, const zk_qap& qap, const uint64_t* weights, size_t m, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out) {
    DevBuf<Fr> dw(std::max<size_t>(m, 1));
    if (m) ZK_HIP(hipMemcpyAsync(dw.p, weights, m * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    prove_dev(ctx, crs, qap, dw.p, m, r, s, proof_out, 0, 1, nullptr);
}

void prove_combine(zk_ctx* ctx, const zk_crs& crs, const void* d_partials, int world, const uint64_t r[4], const uint64_t s[4], uint8_t* proof_out) {
    DevBuf<MsmResults> d_ms(1);
    DevBuf<AssemblePre> d_pre(1);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(320), 0, ctx->stream, (const uint8_t*)d_partials, world, d_ms.p);
    ZK_HIP(hipGetLastError());
    finish(ctx, crs, d_ms.p, r, s, d_pre.p, false, proof_out);
}

}  // namespace zk
