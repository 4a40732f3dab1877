// crs.hip -- device-resident CRS (SigmaG1/SigmaG2, /root/reference/src/groth16/mod.rs:105-121):
// upload, download, the bit-reversed copies used by the roots-of-unity pipeline, and
// groth16::setup (/root/reference/src/groth16/mod.rs:134-197) on the GPU with the trapdoor
// injected (the reference draws it from thread_rng, mod.rs:139-145).
#include <random>
#include "pipeline.hpp"

namespace zk {

// ---- upload / download -------------------------------------------------------------------
template <class A>
static void up_points(zk_ctx* ctx, DevBuf<A>& d, const uint64_t* src, size_t count, int* d_flag) {
    d.alloc(std::max<size_t>(count, 1));
    if (!count) return;
    ZK_REQUIRE(src, ZK_ERR_ARG, "zk_crs_upload: null point array");
    ZK_HIP(hipMemcpyAsync(d.p, src, count * sizeof(A), hipMemcpyHostToDevice, ctx->stream));
    pts_to_mont<A>(ctx, d.p, d.p, count, d_flag);
    pts_check_on_curve<A>(ctx, d.p, count, d_flag);
}

// ---- uploaded G2 points must lie in the r-torsion subgroup --------------------------------------------------
// The twist E'(Fq2) has order r (2q - r) with 2q - r = 10069 . 5864401 . 1875725156269 . (a 181-bit prime): an on-curve point
// with a component of small order d would make B = sum v_k P_k leak v_k mod d (small-subgroup attack on the witness scalars).
// Single points and short arrays are multiplied by r one by one; a long array is checked through two random linear combinations
// sum rho_i P_i (64-bit rho_i, one MSM each) whose result is multiplied by r: a component outside the subgroup survives a
// combination with probability >= 1 - 1 / 10069, so two independent ones miss it with probability < 1e-8.
__global__ void k_random_scalars64(uint64_t seed, Fr* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);     // SplitMix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    Fr k = Fr::zero();
    k.l[0] = (uint32_t)z | 1u;
    k.l[1] = (uint32_t)(z >> 32);
    out[i] = k;
}
// flag |= 8 unless r * P is infinity; one point per lane (Jacobian input: stride 1) or per affine point
template <class P>
__global__ void k_g2_times_r(const P* __restrict__ pts, size_t n, int* flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G2J p;
    if constexpr (sizeof(P) == sizeof(G2A)) { if (pts[i].is_inf()) return; p = G2J::from_affine(reinterpret_cast<const G2A*>(pts)[i]); }
    else p = reinterpret_cast<const G2J*>(pts)[i];
    uint32_t r[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) r[w] = FrParams::P[w];
    if (!jac_mul_words(p, r).is_inf()) atomicOr(flag, 8);
}
static void g2_subgroup_check(zk_ctx* ctx, const G2A* d_pts, size_t n, int* d_flag) {
    if (!n) return;
    if (n <= 4096) {
        hipLaunchKernelGGL(k_g2_times_r<G2A>, dim3(ceil_div(n, 64)), dim3(64), 0, ctx->stream, d_pts, n, d_flag);
        ZK_HIP(hipGetLastError());
        return;
    }
    std::random_device rd;
    MsmTable<Fq2> tab;
    msm_build_table<Fq2>(ctx, d_pts, n, msm_auto_window_g2(n), tab);
    DevBuf<Fr> rho(n);
    DevBuf<G2J> sum(2);
    MsmWorkspace ws;
    for (int trial = 0; trial < 2; ++trial) {
        const uint64_t seed = ((uint64_t)rd() << 32) | rd();
        hipLaunchKernelGGL(k_random_scalars64, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, seed, rho.p, n);
        msm_run<Fq2>(ctx, ws, ctx->stream, tab, rho.p, n, 0, 1, sum.p + trial);
    }
    hipLaunchKernelGGL(k_g2_times_r<G2J>, dim3(1), dim3(64), 0, ctx->stream, sum.p, (size_t)2, d_flag);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(ctx->stream));   // the table and the workspace go out of scope
}

zk_crs* crs_upload(zk_ctx* ctx, const zk_crs_desc& d) {
    ZK_REQUIRE(d.n >= 1 && d.m >= 1 && d.input < d.m, ZK_ERR_ARG, "zk_crs_upload: need n >= 1 and input < m");
    std::unique_ptr<zk_crs> c(new zk_crs());
    c->ctx = ctx;
    c->n = d.n;
    c->m = d.m;
    c->input = d.input;
    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    up_points(ctx, c->alpha1, d.alpha_g1, 1, flag.p);
    up_points(ctx, c->beta1, d.beta_g1, 1, flag.p);
    up_points(ctx, c->delta1, d.delta_g1, 1, flag.p);
    up_points(ctx, c->xi1, d.xi_g1, d.n, flag.p);
    up_points(ctx, c->sum_gamma1, d.sum_gamma_g1, d.input + 1, flag.p);
    up_points(ctx, c->sum_delta1, d.sum_delta_g1, d.m - d.input - 1, flag.p);
    up_points(ctx, c->xi_t1, d.xi_t_g1, d.n - 1, flag.p);
    up_points(ctx, c->beta2, d.beta_g2, 1, flag.p);
    up_points(ctx, c->gamma2, d.gamma_g2, 1, flag.p);
    up_points(ctx, c->delta2, d.delta_g2, 1, flag.p);
    up_points(ctx, c->xi2, d.xi_g2, d.n, flag.p);
    int h = 0;
    ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!(h & 2), ZK_ERR_RANGE, "zk_crs_upload: coordinate >= q");
    ZK_REQUIRE(!(h & 4), ZK_ERR_RANGE, "zk_crs_upload: point not on the curve");
    g2_subgroup_check(ctx, c->beta2.p, 1, flag.p);
    g2_subgroup_check(ctx, c->gamma2.p, 1, flag.p);
    g2_subgroup_check(ctx, c->delta2.p, 1, flag.p);
    g2_subgroup_check(ctx, c->xi2.p, d.n, flag.p);
    ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!(h & 8), ZK_ERR_RANGE, "zk_crs_upload: G2 point outside the subgroup of order r");
    return c.release();
}

void crs_dims(const zk_crs& c, size_t* n, size_t* m, size_t* input) {
    if (n) *n = c.n;
    if (m) *m = c.m;
    if (input) *input = c.input;
}

template <class A>
static void down_points(zk_ctx* ctx, const DevBuf<A>& d, uint64_t* dst, size_t count) {
    if (!dst || !count) return;
    DevBuf<A> tmp(count);
    pts_from_mont<A>(ctx, d.p, tmp.p, count);
    ZK_HIP(hipMemcpyAsync(dst, tmp.p, count * sizeof(A), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}

void crs_download(zk_ctx* ctx, const zk_crs& c, const zk_crs_out& o) {
    down_points(ctx, c.alpha1, o.alpha_g1, 1);
    down_points(ctx, c.beta1, o.beta_g1, 1);
    down_points(ctx, c.delta1, o.delta_g1, 1);
    down_points(ctx, c.xi1, o.xi_g1, c.n);
    down_points(ctx, c.sum_gamma1, o.sum_gamma_g1, c.input + 1);
    down_points(ctx, c.sum_delta1, o.sum_delta_g1, c.m - c.input - 1);
    down_points(ctx, c.xi_t1, o.xi_t_g1, c.n - 1);
    down_points(ctx, c.beta2, o.beta_g2, 1);
    down_points(ctx, c.gamma2, o.gamma_g2, 1);
    down_points(ctx, c.delta2, o.delta_g2, 1);
    down_points(ctx, c.xi2, o.xi_g2, c.n);
}

// The Lagrange-basis arrays of an integer-roots CRS (aproots.hip) -- for the file container: host words <-> device.  They are
// range- and curve-checked like every uploaded point; that they are the same CRS in another basis is the file's business (it is
// the prover's own file, checksummed; a wrong array yields proofs that zk_verify rejects).
void crs_download_lagrange(zk_ctx* ctx, const zk_crs& c, uint64_t* lag1, uint64_t* lagS_t1, uint64_t* lag2) {
    ZK_REQUIRE(c.ap, ZK_ERR_ARG, "CRS holds no Lagrange-basis arrays");
    down_points(ctx, c.lag1, lag1, c.n);
    down_points(ctx, c.lagS_t1, lagS_t1, c.n - 1);
    down_points(ctx, c.lag2, lag2, c.n);
}
void crs_attach_lagrange(zk_ctx* ctx, zk_crs& c, const uint64_t* lag1, const uint64_t* lagS_t1, const uint64_t* lag2) {
    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    up_points(ctx, c.lag1, lag1, c.n, flag.p);
    up_points(ctx, c.lagS_t1, lagS_t1, c.n - 1, flag.p);
    up_points(ctx, c.lag2, lag2, c.n, flag.p);
    int h = 0;
    ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!(h & 2), ZK_ERR_RANGE, "zk_crs_load: coordinate >= q");
    ZK_REQUIRE(!(h & 4), ZK_ERR_RANGE, "zk_crs_load: point not on the curve");
    g2_subgroup_check(ctx, c.lag2.p, c.n, flag.p);
    ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!(h & 8), ZK_ERR_RANGE, "zk_crs_load: G2 point outside the subgroup of order r");
    c.ap = true;
}

void crs_free(zk_crs* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    delete c;
}

// out[brev(i)] = i < count ? in[i] : infinity,  i < 2^log_n
template <class A>
__global__ void k_points_brev(const A* __restrict__ in, size_t count, A* __restrict__ out, unsigned log_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_n)) return;
    uint32_t j = log_n ? (__brev((uint32_t)i) >> (32 - log_n)) : 0;
    out[j] = i < count ? in[i] : A::infinity();
}

void crs_ensure_brev(zk_ctx* ctx, zk_crs& c, unsigned log_n) {
    if (c.has_br && c.br_log_n == log_n) return;
    size_t n = (size_t)1 << log_n;
    ZK_REQUIRE(n == c.n, ZK_ERR_ARG, "CRS degree does not match the QAP domain");
    c.xi1_br.alloc(n);
    c.xi_t1_br.alloc(n);
    c.xi2_br.alloc(n);
    dim3 g(ceil_div(n, 256)), b(256);
    hipLaunchKernelGGL(k_points_brev<G1A>, g, b, 0, ctx->stream, c.xi1.p, n, c.xi1_br.p, log_n);
    hipLaunchKernelGGL(k_points_brev<G1A>, g, b, 0, ctx->stream, c.xi_t1.p, n - 1, c.xi_t1_br.p, log_n);
    hipLaunchKernelGGL(k_points_brev<G2A>, g, b, 0, ctx->stream, c.xi2.p, n, c.xi2_br.p, log_n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    c.has_br = true;
    c.br_log_n = log_n;
}

void crs_ensure_tables(zk_ctx* ctx, zk_crs& c, bool brev, unsigned log_n, bool lagrange) {
    const int kind = lagrange ? 2 : (brev ? 1 : 0);
    if (c.tables_kind == kind && c.tables_c == ctx->opt_window_bits) return;
    ZK_REQUIRE(!lagrange || c.ap, ZK_ERR_UNSUPPORTED, "prove: an integer-roots QAP needs the CRS zk_setup made for it (Lagrange-basis points)");
    if (brev) crs_ensure_brev(ctx, c, log_n);
    // msm_window_bits: 0 = automatic, c = the same window for every table, 100*big + small = `big` for tables of
    // 2^21 points and more and `small` below, + 10000*g2 = its own window for the G2 table (tuning sweeps)
    const long o_all = ctx->opt_window_bits, o_g2 = o_all / 10000;
    auto pick = [&](size_t count) {
        const long o = o_all % 10000;
        if (o <= 0) return msm_auto_window(count);
        if (o < 100) return (int)o;
        return (int)(count >= ((size_t)1 << 21) - 8 ? o / 100 : o % 100);
    };
    const size_t n = c.n, nl = c.m - c.input - 1;
    // integer-roots form: the inner products run over the Lagrange-basis points instead of the powers (natural order)
    const G1A* b_xi1 = lagrange ? c.lag1.p : (brev ? c.xi1_br.p : c.xi1.p);
    const G1A* b_xit = lagrange ? c.lagS_t1.p : (brev ? c.xi_t1_br.p : c.xi_t1.p);
    const G2A* b_xi2 = lagrange ? c.lag2.p : (brev ? c.xi2_br.p : c.xi2.p);
    // xi_t has n-1 points; the bit-reversed copy is padded with infinity to n entries
    msm_build_table<Fq>(ctx, b_xi1, n, pick(n), c.t_xi1);
    {   // bases of the merged product H + r*B1 + s*A + L: xi_t | xi | sum_delta.  Everything the proof element c takes from the witness
        // and from h is ONE inner product over this table (prove.hip): one set of buckets, one reduction tail.  An entry point that
        // needs L on its own (the scalar exchange without per-rank tables, batches) multiplies the points from off_l on.
        const size_t nt = brev ? n : n - 1;
        DevBuf<G1A> cat(nt + n + nl);
        if (nt) ZK_HIP(hipMemcpyAsync(cat.p, b_xit, nt * sizeof(G1A), hipMemcpyDeviceToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(cat.p + nt, b_xi1, n * sizeof(G1A), hipMemcpyDeviceToDevice, ctx->stream));
        if (nl) ZK_HIP(hipMemcpyAsync(cat.p + nt + n, c.sum_delta1.p, nl * sizeof(G1A), hipMemcpyDeviceToDevice, ctx->stream));
        msm_build_table<Fq>(ctx, cat.p, nt + n + nl, pick(nt + n + nl), c.t_hb1);
        c.off_l = nt + n;
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    }
    msm_build_table<Fq2>(ctx, b_xi2, n, o_g2 > 0 ? (int)o_g2 : (o_all % 10000 > 0 ? pick(n) : msm_auto_window_g2(n)), c.t_xi2);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (brev) {   // the tables now hold the permuted points
        c.xi1_br.release(); c.xi_t1_br.release(); c.xi2_br.release();
        c.has_br = false;
    }
    c.tables_kind = kind;
    c.tables_c = ctx->opt_window_bits;
}

// The same tables restricted to the points rank `rank` of `world` owns in the scalar exchange: [rank c, (rank + 1) c) of every
// product, c = cl / cn / ch (ExchangeDims, prove.hip).
void crs_ensure_rank_tables(zk_ctx* ctx, zk_crs& c, bool brev, unsigned log_n, bool lagrange, int rank, int world, size_t cl, size_t cn, size_t ch) {
    const int kind = lagrange ? 2 : (brev ? 1 : 0);
    zk_crs::RankTables& R = c.rank_tabs;
    if (R.rank == rank && R.world == world && R.kind == kind && R.c_opt == ctx->opt_window_bits) return;
    ZK_REQUIRE(!lagrange || c.ap, ZK_ERR_UNSUPPORTED, "prove: an integer-roots QAP needs the CRS zk_setup made for it (Lagrange-basis points)");
    if (R.rank >= 0) ZK_HIP(hipDeviceSynchronize());   // another rank's tables are replaced (one device playing several ranks): nothing may still read them
    if (brev) crs_ensure_brev(ctx, c, log_n);
    const long o_all = ctx->opt_window_bits, o_g2 = o_all / 10000;
    auto pick = [&](size_t count) {
        const long o = o_all % 10000;
        if (o <= 0) return msm_auto_window(count);
        if (o < 100) return (int)o;
        return (int)(count >= ((size_t)1 << 21) - 8 ? o / 100 : o % 100);
    };
    const size_t n = c.n, nl = c.m - c.input - 1, g = (size_t)rank;
    auto range = [&](size_t chunk, size_t count, size_t* lo) {
        *lo = std::min(g * chunk, count);
        return std::min(chunk, count - *lo);
    };
    const G1A* b_xi1 = lagrange ? c.lag1.p : (brev ? c.xi1_br.p : c.xi1.p);
    const G1A* b_xit = lagrange ? c.lagS_t1.p : (brev ? c.xi_t1_br.p : c.xi_t1.p);
    const G2A* b_xi2 = lagrange ? c.lag2.p : (brev ? c.xi2_br.p : c.xi2.p);
    size_t lo, cnt;
    auto build1 = [&](const G1A* pts, size_t count, size_t chunk, MsmTable<Fq>& out) {
        cnt = range(chunk, count, &lo);
        msm_build_table<Fq>(ctx, pts + lo, cnt, pick(cnt), out);   // the window of the rank's own point count: `world` groups share the buckets
    };
    build1(b_xi1, n, cn, R.t_xi1);
    {
        // the rank's points of xi_t | xi ([g ch, (g + 1) ch), padded with infinity to ch entries) followed by its points of sum_delta
        // ([g cl, (g + 1) cl)): ONE table for the merged product L + H of the exchange (prove_msm_submit); sum_delta starts at point ch
        const size_t nt = brev ? n : n - 1;
        DevBuf<G1A> cat(nt + n), mine(ch + cl);
        if (nt) ZK_HIP(hipMemcpyAsync(cat.p, b_xit, nt * sizeof(G1A), hipMemcpyDeviceToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(cat.p + nt, b_xi1, n * sizeof(G1A), hipMemcpyDeviceToDevice, ctx->stream));
        ZK_HIP(hipMemsetAsync(mine.p, 0, (ch + cl) * sizeof(G1A), ctx->stream));   // all-zero = the point at infinity
        size_t lo_h, lo_l;
        const size_t cnt_h = range(ch, nt + n, &lo_h), cnt_l = range(cl, nl, &lo_l);
        if (cnt_h) ZK_HIP(hipMemcpyAsync(mine.p, cat.p + lo_h, cnt_h * sizeof(G1A), hipMemcpyDeviceToDevice, ctx->stream));
        if (cnt_l) ZK_HIP(hipMemcpyAsync(mine.p + ch, c.sum_delta1.p + lo_l, cnt_l * sizeof(G1A), hipMemcpyDeviceToDevice, ctx->stream));
        msm_build_table<Fq>(ctx, mine.p, ch + cl, pick(ch + cl), R.t_hb1);
        R.off_l = ch;
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    }
    cnt = range(cn, n, &lo);
    msm_build_table<Fq2>(ctx, b_xi2 + lo, cnt, o_g2 > 0 ? (int)o_g2 : (o_all % 10000 > 0 ? pick(cnt) : msm_auto_window_g2(cnt)), R.t_xi2);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (brev) {   // the tables hold the permuted points
        c.xi1_br.release(); c.xi_t1_br.release(); c.xi2_br.release();
        c.has_br = false;
    }
    R.rank = rank; R.world = world; R.kind = kind; R.c_opt = ctx->opt_window_bits;
}

// FT[w][d] = d * 16^w * P, w < 64, d < 16 (one lane per window; built once per CRS)
template <class F>
__global__ void k_fixed_table(const Aff<F>* __restrict__ point, Aff<F>* __restrict__ table) {
    int w = threadIdx.x;
    if (w >= 64) return;
    Jac<F> base = Jac<F>::from_affine(*point);
    for (int k = 0; k < 4 * w; ++k) base = jac_dbl_ni(base);
    Jac<F> acc = Jac<F>::infinity();
    for (int d = 0; d < 16; ++d) {
        table[w * 16 + d] = jac_to_affine(acc);
        acc = jac_add_ni(acc, base);
    }
}

void crs_ensure_fixed_tables(zk_ctx* ctx, zk_crs& c) {
    if (c.has_ft) return;
    auto build1 = [&](const DevBuf<G1A>& p, DevBuf<G1A>& t) {
        t.alloc(1024);
        hipLaunchKernelGGL(k_fixed_table<Fq>, dim3(1), dim3(64), 0, ctx->stream, p.p, t.p);
    };
    auto build2 = [&](const DevBuf<G2A>& p, DevBuf<G2A>& t) {
        t.alloc(1024);
        hipLaunchKernelGGL(k_fixed_table<Fq2>, dim3(1), dim3(64), 0, ctx->stream, p.p, t.p);
    };
    build1(c.alpha1, c.ft_alpha1);
    build1(c.beta1, c.ft_beta1);
    build1(c.delta1, c.ft_delta1);
    build2(c.beta2, c.ft_beta2);
    build2(c.delta2, c.ft_delta2);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    c.has_ft = true;
}

// ---- setup ---------------------------------------------------------------------------------
struct SetupConsts {
    Fr alpha, beta, gamma, delta, x;
    Fr gamma_inv, delta_inv;
    Fr tx;        // t(x)
    Fr tx_dinv;   // t(x) / delta
    Fr lag_c;     // (x^n - 1) / n   (sparse form)
    G1A g1;       // 69 * G1::one()   (fr.rs:106-109)
    G2A g2;       // 96 * G2::one()   (fr.rs:110-113)
};

struct G2GenWords {
    uint32_t x0[8], x1[8], y0[8], y1[8];
};
static const G2GenWords G2GEN = {
    {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu},
    {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u},
    {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u},
    {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u}};

__device__ __forceinline__ Fq fq_from_words(const uint32_t* w) {
    Fq x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x.l[i] = w[i];
    return Fq::from_canonical(x);
}

// one lane: trapdoor (canonical) -> Montgomery constants, inverses, t(x), encryption bases
__global__ void k_setup_consts(const Fr* __restrict__ td, G2GenWords gen, int dense, size_t n, unsigned log_n,
                               const Fr* __restrict__ t_coeffs, Fr tx_given, SetupConsts* __restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    SetupConsts c;
    c.alpha = Fr::from_canonical(td[0]);
    c.beta = Fr::from_canonical(td[1]);
    c.gamma = Fr::from_canonical(td[2]);
    c.delta = Fr::from_canonical(td[3]);
    c.x = Fr::from_canonical(td[4]);
    c.gamma_inv = c.gamma.inv();
    c.delta_inv = c.delta.inv();
    if (dense == 2) {          // integer roots: t(x) = prod (x - k) computed by the host (aproots.hip)
        c.tx = tx_given;
        c.lag_c = Fr::zero();
    } else if (dense) {
        Fr acc = Fr::zero();  // Horner, Polynomial::evaluate (field/mod.rs:338-343)
        for (size_t k = n + 1; k-- > 0;) acc = acc * c.x + t_coeffs[k];
        c.tx = acc;
        c.lag_c = Fr::zero();
    } else {
        Fr xn = c.x;
        for (unsigned k = 0; k < log_n; ++k) xn = xn.sqr();
        c.tx = xn - Fr::one();
        Fr nn = Fr::zero();
        nn.l[0] = (uint32_t)n;
        c.lag_c = c.tx * Fr::from_canonical(nn).inv();
    }
    c.tx_dinv = c.tx * c.delta_inv;
    G1J one1{Fq::from_u32(1), Fq::from_u32(2), Fq::one()};
    c.g1 = jac_to_affine(jac_mul_small(one1, 69));
    G2J one2{Fq2{fq_from_words(gen.x0), fq_from_words(gen.x1)}, Fq2{fq_from_words(gen.y0), fq_from_words(gen.y1)}, Fq2::one()};
    c.g2 = jac_to_affine(jac_mul_small(one2, 96));
    *out = c;
}

__device__ __forceinline__ Fr pow_u32(Fr base, uint32_t e) {
    Fr acc = Fr::one();
    for (int i = 31 - __clz(e | 1); i >= 0; --i) {
        acc = acc.sqr();
        if ((e >> i) & 1) acc = acc * base;
    }
    return acc;
}

// xi_s[i] = x^i ; xit_s[i] = x^i * t(x)/delta
__global__ void k_setup_powers(const SetupConsts* __restrict__ cs, Fr* __restrict__ xi_s, Fr* __restrict__ xit_s, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr p = pow_u32(cs->x, (uint32_t)i);
    xi_s[i] = p;
    if (i + 1 < n) xit_s[i] = p * cs->tx_dinv;
}

// L[j] = (x^n - 1)/n * w^j / (x - w^j);  if x == w^j then L[j] = 1 (and lag_c == 0 zeroes the rest)
__global__ void k_lagrange_at(const SetupConsts* __restrict__ cs, Fr w, Fr* __restrict__ L, size_t n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Fr wj = pow_u32(w, (uint32_t)j);
    Fr den = cs->x - wj;
    L[j] = den.is_zero() ? Fr::one() : cs->lag_c * wj * den.inv();
}

// comb[i] = (beta*u_i(x) + alpha*v_i(x) + w_i(x)) / (i <= l ? gamma : delta)   (mod.rs:147-164)
// One workgroup per COMB_ROWS wires; a wire's entries are walked by the whole wave it belongs to when it has
// many of them (a wire that feeds every gate, like x in the chain circuit, has n entries: one lane took a second
// for it at 2^20), so rows are processed wave-cooperatively: 64 lanes stride over the entries, then a shuffle tree.
__device__ __forceinline__ Fr wave_sum(Fr v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Fr o;
#pragma unroll
        for (int l = 0; l < 8; ++l) o.l[l] = __shfl_down(v.l[l], off);
        v = v + o;
    }
    return v;   // valid in lane 0
}
__global__ __launch_bounds__(256) void k_setup_comb_sparse(const SetupConsts* __restrict__ cs, const Fr* __restrict__ L,
                                    const uint32_t* up, const uint32_t* ug, const Fr* uv,
                                    const uint32_t* vp, const uint32_t* vg, const Fr* vv,
                                    const uint32_t* wp, const uint32_t* wg, const Fr* wv,
                                    size_t m, size_t input, Fr* __restrict__ comb) {
    // one wave per wire
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= m) return;
    Fr ux = Fr::zero(), vx = Fr::zero(), wx = Fr::zero();
    for (uint32_t k = up[i] + lane; k < up[i + 1]; k += 64) ux = ux + uv[k] * L[ug[k]];
    for (uint32_t k = vp[i] + lane; k < vp[i + 1]; k += 64) vx = vx + vv[k] * L[vg[k]];
    for (uint32_t k = wp[i] + lane; k < wp[i + 1]; k += 64) wx = wx + wv[k] * L[wg[k]];
    ux = wave_sum(ux); vx = wave_sum(vx); wx = wave_sum(wx);
    if (lane == 0) {
        Fr c = cs->beta * ux + cs->alpha * vx + wx;
        comb[i] = c * (i <= input ? cs->gamma_inv : cs->delta_inv);
    }
}
__global__ void k_setup_comb_dense(const SetupConsts* __restrict__ cs, const Fr* __restrict__ U, const Fr* __restrict__ V,
                                   const Fr* __restrict__ W, size_t m, size_t n, size_t input, Fr* __restrict__ comb) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fr x = cs->x, ux = Fr::zero(), vx = Fr::zero(), wx = Fr::zero();
    for (size_t k = n; k-- > 0;) {
        ux = ux * x + U[i * n + k];
        vx = vx * x + V[i * n + k];
        wx = wx * x + W[i * n + k];
    }
    Fr c = cs->beta * ux + cs->alpha * vx + wx;
    comb[i] = c * (i <= input ? cs->gamma_inv : cs->delta_inv);
}

// out[i] = scalars[i] * base  (encrypt_g1 / encrypt_g2, fr.rs:106-113), one lane per scalar, through the 4-bit
// table FT[w][d] = d 16^w base (k_fixed_table): at most 64 mixed additions instead of 254 doublings + ~127 additions
template <class F>
__global__ __launch_bounds__(64) void k_fixed_base_mul(const Aff<F>* __restrict__ table, const Fr* __restrict__ scalars, Aff<F>* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k = scalars[i].to_canonical();
    Jac<F> acc = Jac<F>::infinity();
    for (int w = 0; w < 64; ++w) {
        const uint32_t d = (k.l[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (d) acc = jac_madd_ni(acc, table[w * 16 + d]);
    }
    out[i] = jac_to_affine(acc);
}
template <class F>
static void fixed_base_mul(zk_ctx* ctx, const Aff<F>* table, const Fr* scalars, Aff<F>* out, size_t n, const char* name) {
    if (!n) return;
    ProfScope ps(ctx, name, (32.0 + sizeof(Aff<F>)) * n);
    hipLaunchKernelGGL(k_fixed_base_mul<F>, dim3(ceil_div(n, 64)), dim3(64), 0, ctx->stream, table, scalars, out, n);
    ZK_HIP(hipGetLastError());
}

zk_crs* crs_setup(zk_ctx* ctx, const zk_qap& q, const uint64_t trapdoor[20]) {
    for (int k = 0; k < 5; ++k) {
        const uint64_t* e = trapdoor + 4 * k;
        // Random for FrLocal never yields 0 (fr.rs:90-99); gamma/delta == 0 would panic in `/` (fr.rs:54)
        ZK_REQUIRE(e[0] | e[1] | e[2] | e[3], ZK_ERR_DIV_BY_ZERO, "zk_setup: trapdoor elements must be non-zero");
    }
    const size_t n = q.n, m = q.m, l = q.input;
    std::unique_ptr<zk_crs> c(new zk_crs());
    c->ctx = ctx;
    c->n = n;
    c->m = m;
    c->input = l;
    hipStream_t st = ctx->stream;
    DevBuf<Fr> td(5);
    DevBuf<int> flag(1);
    DevBuf<SetupConsts> cs(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    ZK_HIP(hipMemcpyAsync(td.p, trapdoor, 5 * sizeof(Fr), hipMemcpyHostToDevice, st));
    {
        DevBuf<Fr> tmp(5);
        fr_to_mont(ctx, td.p, tmp.p, 5, flag.p);  // range check only
        int h = 0;
        ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        ZK_REQUIRE(!h, ZK_ERR_RANGE, "zk_setup: trapdoor element >= r");
    }
    const bool ap = !q.dense && q.roots == 1, arb = !q.dense && q.roots == 2;
    hipLaunchKernelGGL(k_setup_consts, dim3(1), dim3(64), 0, st, td.p, G2GEN, (ap || arb) ? 2 : (q.dense ? 1 : 0), n, q.log_n, q.dt.p,
                       ap ? ap_t_at_x(q, trapdoor) : arb ? arb_t_at_x(q, trapdoor) : Fr::zero(), cs.p);
    ZK_HIP(hipGetLastError());
    DevBuf<Fr> apL(ap ? n : 0), apLS(ap ? std::max<size_t>(n - 1, 1) : 0);

    DevBuf<Fr> xi_s(n), xit_s(std::max<size_t>(n, 1)), comb(m);
    hipLaunchKernelGGL(k_setup_powers, dim3(ceil_div(n, 256)), dim3(256), 0, st, cs.p, xi_s.p, xit_s.p, n);
    if (q.dense) {
        hipLaunchKernelGGL(k_setup_comb_dense, dim3(ceil_div(m, 64)), dim3(64), 0, st, cs.p, q.du.p, q.dv.p, q.dw.p, m, n, l, comb.p);
    } else if (ap) {
        ap_setup_lagrange(ctx, q, trapdoor, apL.p, apLS.p, flag.p);
        hipLaunchKernelGGL(k_setup_comb_sparse, dim3(ceil_div(m * 64, 256)), dim3(256), 0, st, cs.p, apL.p,
                           q.u_wire.ptr.p, q.u_wire.idx.p, q.u_wire.val.p, q.v_wire.ptr.p, q.v_wire.idx.p, q.v_wire.val.p,
                           q.w_wire.ptr.p, q.w_wire.idx.p, q.w_wire.val.p, m, l, comb.p);
        ZK_HIP(hipGetLastError());
        int h = 0;
        ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        ZK_REQUIRE(!(h & 8), ZK_ERR_UNSUPPORTED, "zk_setup: the trapdoor's x is one of the integers 1..2n-1 (draw another)");
    } else if (arb) {
        DevBuf<Fr> L(n);
        arb_setup_lagrange(ctx, q, trapdoor, L.p, flag.p);
        hipLaunchKernelGGL(k_setup_comb_sparse, dim3(ceil_div(m * 64, 256)), dim3(256), 0, st, cs.p, L.p,
                           q.u_wire.ptr.p, q.u_wire.idx.p, q.u_wire.val.p, q.v_wire.ptr.p, q.v_wire.idx.p, q.v_wire.val.p,
                           q.w_wire.ptr.p, q.w_wire.idx.p, q.w_wire.val.p, m, l, comb.p);
        ZK_HIP(hipGetLastError());
        int h = 0;
        ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        ZK_REQUIRE(!(h & 8), ZK_ERR_UNSUPPORTED, "zk_setup: the trapdoor's x is one of the QAP's roots (draw another)");
    } else {
        DevBuf<Fr> L(n);
        hipLaunchKernelGGL(k_lagrange_at, dim3(ceil_div(n, 256)), dim3(256), 0, st, cs.p, host_root_of_unity(q.log_n), L.p, n);
        hipLaunchKernelGGL(k_setup_comb_sparse, dim3(ceil_div(m * 64, 256)), dim3(256), 0, st, cs.p, L.p,
                           q.u_wire.ptr.p, q.u_wire.idx.p, q.u_wire.val.p, q.v_wire.ptr.p, q.v_wire.idx.p, q.v_wire.val.p,
                           q.w_wire.ptr.p, q.w_wire.idx.p, q.w_wire.val.p, m, l, comb.p);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(st));  // L is freed at scope exit
    }
    ZK_HIP(hipGetLastError());

    // 4-bit window tables of the two encryption bases
    DevBuf<G1A> ft1(1024);
    DevBuf<G2A> ft2(1024);
    hipLaunchKernelGGL(k_fixed_table<Fq>, dim3(1), dim3(64), 0, st, &cs.p->g1, ft1.p);
    hipLaunchKernelGGL(k_fixed_table<Fq2>, dim3(1), dim3(64), 0, st, &cs.p->g2, ft2.p);
    ZK_HIP(hipGetLastError());
    const G1A* g1 = ft1.p;
    const G2A* g2 = ft2.p;
    c->xi1.alloc(n);
    c->xi2.alloc(n);
    c->xi_t1.alloc(std::max<size_t>(n - 1, 1));
    c->sum_gamma1.alloc(l + 1);
    c->sum_delta1.alloc(std::max<size_t>(m - l - 1, 1));
    c->alpha1.alloc(1); c->beta1.alloc(1); c->delta1.alloc(1);
    c->beta2.alloc(1); c->gamma2.alloc(1); c->delta2.alloc(1);
    fixed_base_mul<Fq>(ctx, g1, xi_s.p, c->xi1.p, n, "setup_fixed_base_g1");
    fixed_base_mul<Fq>(ctx, g1, xit_s.p, c->xi_t1.p, n - 1, "setup_fixed_base_g1");
    fixed_base_mul<Fq>(ctx, g1, comb.p, c->sum_gamma1.p, l + 1, "setup_fixed_base_g1");
    fixed_base_mul<Fq>(ctx, g1, comb.p + l + 1, c->sum_delta1.p, m - l - 1, "setup_fixed_base_g1");
    fixed_base_mul<Fq2>(ctx, g2, xi_s.p, c->xi2.p, n, "setup_fixed_base_g2");
    if (ap) {   // the same CRS in the Lagrange bases of R and S (see aproots.hip)
        c->ap = true;
        c->lag1.alloc(n); c->lag2.alloc(n); c->lagS_t1.alloc(std::max<size_t>(n - 1, 1));
        fixed_base_mul<Fq>(ctx, g1, apL.p, c->lag1.p, n, "setup_fixed_base_g1");
        fixed_base_mul<Fq2>(ctx, g2, apL.p, c->lag2.p, n, "setup_fixed_base_g2");
        fixed_base_mul<Fq>(ctx, g1, apLS.p, c->lagS_t1.p, n - 1, "setup_fixed_base_g1");
    }
    // alpha, beta, gamma, delta sit first in SetupConsts (Montgomery)
    const Fr* tdm = &cs.p->alpha;
    fixed_base_mul<Fq>(ctx, g1, tdm + 0, c->alpha1.p, 1, "setup_fixed_base_g1");
    fixed_base_mul<Fq>(ctx, g1, tdm + 1, c->beta1.p, 1, "setup_fixed_base_g1");
    fixed_base_mul<Fq>(ctx, g1, tdm + 3, c->delta1.p, 1, "setup_fixed_base_g1");
    fixed_base_mul<Fq2>(ctx, g2, tdm + 1, c->beta2.p, 1, "setup_fixed_base_g2");
    fixed_base_mul<Fq2>(ctx, g2, tdm + 2, c->gamma2.p, 1, "setup_fixed_base_g2");
    fixed_base_mul<Fq2>(ctx, g2, tdm + 3, c->delta2.p, 1, "setup_fixed_base_g2");
    ZK_HIP(hipStreamSynchronize(st));
    ctx->resolve_profile();
    return c.release();
}

}  // namespace zk
