// ntt.hip -- radix-2 NTT over the BN254 scalar field for gfx950.
//
// Replaces the O(n^2) polynomial algebra on groth16::prove's path -- CoefficientPoly Mul
// (/root/reference/src/groth16/coefficient_poly.rs:93-130), Div -> polynomial_division
// (/root/reference/src/field/mod.rs:428-469) and the Lagrange interpolation of
// QAP::from (/root/reference/src/groth16/fr.rs:140-173) -- with transforms whose semantics
// are exactly field::dft / field::idft (/root/reference/src/field/mod.rs:508-537).
//
// Structure (MI355X-first): a transform of size n = 2^log_n is at most two passes over HBM.
// Each workgroup stages a 2048-element tile (64 KiB, limb-planar so every ds_read_b32 of
// consecutive lanes is conflict free) in LDS, runs up to 11 butterfly stages on it, and writes it
// back.  The column pass gathers rows of `cols` contiguous elements (>= 128 B at n = 2^20) so HBM
// reads stay coalesced; the row pass is fully contiguous.  Decimation-in-frequency maps natural
// -> bit-reversed order and decimation-in-time maps bit-reversed -> natural, so the prove
// pipeline never runs a separate bit-reversal pass (the CRS is stored in matching order).
// The inter-pass twiddles, the coset factors and 1/n are fused into the tile load/store.
#include "kernels.hpp"
#include "lazy29.cuh"
#include "fr_tile.cuh"

namespace zk {

// ---- host-side constants -----------------------------------------------------------------
Fr host_fr_from_u64(uint64_t v) {
    Fr x = Fr::zero();
    x.l[0] = (uint32_t)v;
    x.l[1] = (uint32_t)(v >> 32);
    return Fr::from_canonical(x);
}
Fr host_fr_pow(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    for (int i = 63; i >= 0; --i) {
        acc = acc.sqr();
        if ((e >> i) & 1) acc = acc * base;
    }
    return acc;
}
Fr host_root_of_unity(unsigned log_n) {
    // 5^((r-1)/2^28), canonical little-endian words
    Fr w;
    const uint32_t W28[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu, 0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
    for (int i = 0; i < 8; ++i) w.l[i] = W28[i];
    w = Fr::from_canonical(w);
    for (unsigned i = log_n; i < 28; ++i) w = w.sqr();
    return w;
}

__device__ __forceinline__ Fr fr_pow_u32(Fr base, uint32_t e) {
    Fr acc = Fr::one();
    for (int i = 31 - __clz(e | 1); i >= 0; --i) {
        acc = acc.sqr();
        if ((e >> i) & 1) acc = acc * base;
    }
    return acc;
}

__device__ __forceinline__ uint32_t brev(uint32_t x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// ---- table generation --------------------------------------------------------------------
__global__ void k_powers(Fr base, Fr scale, Fr* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fr_pow_u32(base, (uint32_t)i) * scale;
}
// out[pos] = base^brev(pos) * scale
__global__ void k_powers_brev(Fr base, Fr scale, Fr* __restrict__ out, unsigned log_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << log_n)) out[i] = fr_pow_u32(base, brev((uint32_t)i, log_n)) * scale;
}
// mid[p * R2 + c] = w^(c * brev_a(p)),  n = R1 * R2, R1 = 2^a
__global__ void k_mid_table(Fr w, Fr* __restrict__ out, unsigned a, unsigned log_r2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << (a + log_r2))) return;
    uint32_t p = (uint32_t)(i >> log_r2), c = (uint32_t)(i & (((size_t)1 << log_r2) - 1));
    out[i] = fr_pow_u32(w, c * brev(p, a));
}

void fr_powers(zk_ctx* ctx, Fr base, Fr scale, Fr* out, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_powers, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, base, scale, out, n);
    ZK_HIP(hipGetLastError());
}

static Fr host_inv_pow2(unsigned k) {  // 2^-k
    Fr two = host_fr_from_u64(2);
    return host_fr_pow(two, k).inv();
}

// twiddle of the local butterflies in the tile's own radix: 9 x 29-bit limbs, 12 words apart (three 16-byte loads, no conversion)
constexpr int TW29_STRIDE = 12;
__global__ void k_tw29(const Fr* __restrict__ tw, int count, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const FpR<FrParams> v = FpR<FrParams>::load(tw[i]);
#pragma unroll
    for (int l = 0; l < 9; ++l) out[i * TW29_STRIDE + l] = v.v[l];
#pragma unroll
    for (int l = 9; l < TW29_STRIDE; ++l) out[i * TW29_STRIDE + l] = 0;
}

std::shared_ptr<NttTables> ntt_get_tables(zk_ctx* ctx, unsigned log_n) {
    auto it = ctx->ntt_tables.find(log_n);
    if (it != ctx->ntt_tables.end()) return it->second;
    ZK_REQUIRE(log_n <= NTT_MAX_LOG, ZK_ERR_SIZE, "NTT size above 2^24 is not supported");
    auto t = std::make_shared<NttTables>();
    t->log_n = log_n;
    Fr w2048 = host_root_of_unity(NTT_MAX_LOCAL_LOG);
    t->tw_fwd.alloc(1024);
    t->tw_inv.alloc(1024);
    fr_powers(ctx, w2048, Fr::one(), t->tw_fwd.p, 1024);
    fr_powers(ctx, w2048.inv(), Fr::one(), t->tw_inv.p, 1024);
    t->tw29_fwd.alloc(1024 * TW29_STRIDE);
    t->tw29_inv.alloc(1024 * TW29_STRIDE);
    hipLaunchKernelGGL(k_tw29, dim3(4), dim3(256), 0, ctx->stream, t->tw_fwd.p, 1024, t->tw29_fwd.p);
    hipLaunchKernelGGL(k_tw29, dim3(4), dim3(256), 0, ctx->stream, t->tw_inv.p, 1024, t->tw29_inv.p);
    ZK_HIP(hipGetLastError());
    if (log_n > NTT_MAX_LOCAL_LOG) {
        size_t n = (size_t)1 << log_n;
        // two passes: n = 2^a columns-transform x rows of 2^11; three passes (log_n > 22): 2^(log_n - 22) x blocks of 2^22
        const unsigned log_r2 = log_n > 2 * NTT_MAX_LOCAL_LOG ? 2 * NTT_MAX_LOCAL_LOG : NTT_MAX_LOCAL_LOG;
        unsigned a = log_n - log_r2;
        Fr w = host_root_of_unity(log_n);
        t->mid_fwd.alloc(n);
        t->mid_inv.alloc(n);
        hipLaunchKernelGGL(k_mid_table, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, w, t->mid_fwd.p, a, log_r2);
        hipLaunchKernelGGL(k_mid_table, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, w.inv(), t->mid_inv.p, a, log_r2);
        ZK_HIP(hipGetLastError());
    }
    t->n_inv = host_inv_pow2(log_n);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ntt_tables[log_n] = t;
    return t;
}

void ntt_ensure_coset_tables(zk_ctx* ctx, NttTables& t) {
    if (t.coset_fwd_brev.p) return;
    size_t n = (size_t)1 << t.log_n;
    Fr g = host_root_of_unity(t.log_n + 1);
    t.coset_fwd_brev.alloc(n);
    t.coset_inv_brev_half.alloc(n);
    // both tables carry the 1 / n of the inverse transform in front of them (prove.hip runs those transforms unscaled)
    hipLaunchKernelGGL(k_powers_brev, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, g, t.n_inv, t.coset_fwd_brev.p, t.log_n);
    // ... the second as plain integers (scale / R as a Montgomery-form factor): k_h_combine's product with a Montgomery-form value is canonical
    hipLaunchKernelGGL(k_powers_brev, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, g.inv(), (host_inv_pow2(1) * t.n_inv).to_canonical(), t.coset_inv_brev_half.p, t.log_n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}

// ---- the tile kernel ---------------------------------------------------------------------
// Inside a tile the elements live in the multiplier's own radix (lazy29.cuh: 9 signed 29-bit limbs,
// not reduced): a butterfly's add / sub are 9 independent 32-bit operations instead of two carry
// chains with a conditional correction, and its multiply needs no 8x32 <-> 9x29 conversion.  Each
// lane holds 2^K elements and runs K butterfly stages on them in registers between two visits to
// LDS.  Shipped: K = 2 with 512-lane workgroups (4 elements per lane, 92 VGPRs, 4 waves/SIMD; 11 stages
// = 2+2+2+2+2+1 rounds); K = 3 with 256 lanes (8 elements per lane, 245 VGPRs in the DIT kernel, 2
// waves/SIMD because a 75 KB tile allows two workgroups per CU either way) measured 7 % slower
// stand-alone.  Elements are converted on the tile load and brought back to the canonical 8 x 32 form
// on the tile store, so HBM never sees the lazy form.
//
// Bounds (p = r, "fresh" = a Montgomery output, value in (-0.3p, 1.3p), limbs in normal form):
//  * DIF (x, y) -> (x + y, (x - y) w): the sum path doubles per stage.  Sums are re-normalised
//    (carry propagation) where two un-normalised sums would meet, element 0 of every round -- the only
//    one that is a sum of sums over the whole round -- is reduced modulo p (fr_reduce), so that a
//    round's inputs are below 5.6p, its values below 45p (top limb < 2^28) and every multiplicand
//    x - y has |limb| < 2^30, which is what FpR::mont needs for its 64-bit columns.
//  * DIT (x, y) -> (x + w y, x - w y): values grow by at most 1.3p per stage (< 16p over the 11
//    stages of a tile); limbs are re-normalised before the third multiplication of a round and at
//    its end.
constexpr int TILE = 1 << NTT_MAX_LOCAL_LOG;  // elements per LDS tile
#ifndef ZK_NTT_THREADS
#define ZK_NTT_THREADS 512
#endif
#ifndef ZK_NTT_KMAX
#define ZK_NTT_KMAX 2
#endif
constexpr int NTT_THREADS = ZK_NTT_THREADS;
constexpr int LDS_PLANE = TILE + (TILE >> 6);  // limb-planar, one pad word per 64 elements (stride-8 rounds stay conflict free)
constexpr size_t NTT_LDS_BYTES = (size_t)LDS_PLANE * 9 * sizeof(int32_t);


struct NttPass {
    unsigned log_rows, log_cols;
    unsigned log_tiles;   // tiles per transform; workgroup b * tiles + t is tile t of transform b (batches: data = [batch][n])
    size_t n;             // transform length
    size_t row_stride, tile_stride;
    const int32_t* tw;   // w_2048^(+-k), k < 1024, in the lazy radix (k_tw29)
    const Fr* mid;   // applied on store (data layout) or nullptr
    const Fr* pre;   // applied on load (data layout) or nullptr
    Fr post;
    int has_post;
    int contig;      // a tile = 2^log_cols WHOLE transforms of 2^log_rows contiguous elements each (batches of small transforms): element idx of the tile at base + idx
    size_t pre_step; // > 0: `pre` is ONE table for every transform of the batch -- element i of a transform is multiplied by pre[(i & pre_mask) * pre_step]
    size_t pre_mask; //      (interp.hip: the twist w_4s^i in front of the second half of a parent's image)
    // Fused element-wise work of the prove pipeline (ntt_dif_fused; all null / 0 otherwise).  Transform t of the batch belongs to half
    // sel = t >= fuse_half and is number tl = t - sel fuse_half of it:
    //   load   element from src_a[sel][tl n + ...] instead of the data array, times src_b[sel][tl n + ...] when that is set
    //          (U.V on <w> and on the coset: the point-wise products never exist as arrays);
    //   store  (to the data array as always -- with src_a set the pass reads one array and writes another -- and) ALSO
    //          canonical(value * canon_k) to canon_out[sel][tl n + ...] (the scalars of the A and B products leave the last pass of
    //          the inverse transform directly; canon_k is a plain integer, see qap.hip).
    const Fr* src_a[2];
    const Fr* src_b[2];
    Fr* canon_out[2];
    Fr canon_k;
    unsigned fuse_half;
};

__device__ __forceinline__ FrL lds_get(const int32_t* lds, int e) {
    const int a = e + (e >> 6);
    FrL r;
#pragma unroll
    for (int l = 0; l < 9; ++l) r.v[l] = lds[l * LDS_PLANE + a];
    return r;
}
__device__ __forceinline__ void lds_put(int32_t* lds, int e, const FrL& v) {
    const int a = e + (e >> 6);
#pragma unroll
    for (int l = 0; l < 9; ++l) lds[l * LDS_PLANE + a] = v.v[l];
}

// K butterfly stages (stage numbers s .. s+K-1 of a 2^L-point transform) on the 2^K elements of one unit
__device__ __forceinline__ FrL tw_get(const int32_t* __restrict__ tw, size_t idx) {
    const int4* p = reinterpret_cast<const int4*>(tw + idx * TW29_STRIDE);
    const int4 a = p[0], b = p[1], c = p[2];
    FrL r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w; r.v[8] = c.x;
    return r;
}

// Butterflies whose twiddle is w^0 = 1 are not multiplied.  They sit where the stride is one: in the LAST round of a DIF tile
// (qshift = 0, so j = 0 and pos = q & (half - 1)) and in the FIRST round of a DIT tile (s = 0) -- three of the four butterflies a
// lane runs there, 1.5 of a tile's 9..11 stages; the test is uniform over the workgroup.
template <bool DIT, int K>
__device__ __forceinline__ void ntt_round(int32_t* lds, const int32_t* __restrict__ tw, int col_base, unsigned L, unsigned s, int u, unsigned tw_shift) {
    constexpr int Q = 1 << K;
    FrL x[Q];
    int row0, qshift, j;
    if (!DIT) {
        qshift = (int)(L - s) - K;                 // element q sits at row0 + (q << qshift)
        j = u & ((1 << qshift) - 1);
        row0 = ((u >> qshift) << (L - s)) | j;
    } else {
        qshift = (int)s;
        j = u & ((1 << s) - 1);
        row0 = ((u >> s) << (s + K)) | j;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) x[q] = lds_get(lds, col_base + row0 + (q << qshift));
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const int half = DIT ? (1 << t) : (1 << (K - 1 - t));
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (q & half) continue;
            const bool one = qshift == 0 && (q & (half - 1)) == 0;     // twiddle w^0 (qshift == 0 implies j == 0)
            if (!DIT) {
                // decimation in frequency: (x, y) -> (x + y, (x - y) w),  w = w_{2^(L-s-t)}^pos
                const int pos = ((q & (half - 1)) << qshift) | j;
                FrL sum = x[q] + x[q + half];
                if (t == 1) sum = sum.norm();
                if (one) x[q + half] = (x[q] - x[q + half]).norm();
                else x[q + half] = (x[q] - x[q + half]) * tw_get(tw, ((size_t)pos << (s + t)) << tw_shift);
                x[q] = sum;
            } else {
                // decimation in time: (x, y) -> (x + w y, x - w y),  w = w_{2^(s+t+1)}^pos
                const int pos = ((q & (half - 1)) << qshift) | j;
                const FrL yin = t == 2 ? x[q + half].norm() : x[q + half];
                const FrL y = one ? yin : yin * tw_get(tw, ((size_t)pos << (L - 1 - s - t)) << tw_shift);
                x[q + half] = x[q] - y;
                x[q] = x[q] + y;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (DIT) x[q] = x[q].norm();
        else if (q == 0) x[q] = fr_reduce(x[q]);
        else if (K == 3 && !(q & 1)) x[q] = x[q].norm();
        lds_put(lds, col_base + row0 + (q << qshift), x[q]);
    }
}

template <bool DIT>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_tile(Fr* __restrict__ data, NttPass p) {
    ZK_LATENCY_KERNEL();
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    const unsigned log_rows = p.log_rows, log_cols = p.log_cols;
    const int cols = 1 << log_cols;
    const int elems = 1 << (log_rows + log_cols);
    const unsigned tile = blockIdx.x & ((1u << p.log_tiles) - 1);
    Fr* base = data + (size_t)(blockIdx.x >> p.log_tiles) * p.n + (size_t)tile * p.tile_stride;
    const Fr* pre = p.pre ? (p.pre_step ? p.pre : p.pre + (size_t)tile * p.tile_stride) : nullptr;
    const Fr* mid = p.mid ? p.mid + (size_t)tile * p.tile_stride : nullptr;
    // fused sources / sinks (ntt_dif_fused): uniform over the workgroup
    const unsigned tnum = blockIdx.x >> p.log_tiles, sel = tnum >= p.fuse_half ? 1u : 0u, tl = tnum - sel * p.fuse_half;
    const size_t toff = (size_t)tl * p.n + (size_t)tile * p.tile_stride;
    const Fr* lsrc = p.src_a[sel] ? p.src_a[sel] + toff : base;
    const Fr* lmul = p.src_b[sel] ? p.src_b[sel] + toff : nullptr;
    Fr* scan = p.canon_out[sel] ? p.canon_out[sel] + toff : nullptr;

    // load: consecutive lanes walk the `cols` contiguous elements of a row, then the next row
    for (int idx = threadIdx.x; idx < elems; idx += NTT_THREADS) {
        int row = idx >> log_cols, col = idx & (cols - 1);
        size_t g = (size_t)row * p.row_stride + col;
        int at = (col << log_rows) + row;
        if (p.contig) { g = idx; at = idx; }
        FrL v = FrL::load(lsrc[g]);
        if (lmul) v = v * FrL::load(lmul[g]);
        if (pre) v = v * FrL::load(pre[p.pre_step ? (((size_t)tile * p.tile_stride + g) & p.pre_mask) * p.pre_step : g]);
        lds_put(lds, at, v);
    }
    __syncthreads();

    const unsigned tw_shift = NTT_MAX_LOCAL_LOG - log_rows;
    for (unsigned s = 0; s < log_rows;) {
        const unsigned left = log_rows - s;
        const unsigned k = ZK_NTT_KMAX >= 3 && left >= 3 && left != 4 ? 3 : (left >= 2 ? 2 : 1);   // 4 = 2 + 2
        const int units = elems >> k;
        const unsigned log_upc = log_rows - k;   // units per column
        for (int u = threadIdx.x; u < units; u += NTT_THREADS) {
            const int col_base = (u >> log_upc) << log_rows, uu = u & ((1 << log_upc) - 1);
            if (k == 3) ntt_round<DIT, 3>(lds, p.tw, col_base, log_rows, s, uu, tw_shift);
            else if (k == 2) ntt_round<DIT, 2>(lds, p.tw, col_base, log_rows, s, uu, tw_shift);
            else ntt_round<DIT, 1>(lds, p.tw, col_base, log_rows, s, uu, tw_shift);
        }
        s += k;
        __syncthreads();
    }

    const FrL post = FrL::load(p.post), canon_k = FrL::load(p.canon_k);
    for (int idx = threadIdx.x; idx < elems; idx += NTT_THREADS) {
        int row = idx >> log_cols, col = idx & (cols - 1);
        size_t g = (size_t)row * p.row_stride + col;
        int at = (col << log_rows) + row;
        if (p.contig) { g = idx; at = idx; }
        FrL v = lds_get(lds, at);
        if (mid) v = v * FrL::load(mid[g]);
        if (p.has_post) v = v * post;
        base[g] = fr_store_exact(v);
        if (scan) scan[g] = fr_store_exact(v * canon_k);
    }
}

static void launch_pass(zk_ctx* ctx, bool dit, Fr* d, const NttPass& p, size_t tiles, const char* name, double bytes) {
    ProfScope ps(ctx, name, bytes);
    size_t lds_bytes = NTT_LDS_BYTES;
    if (dit) hipLaunchKernelGGL(k_ntt_tile<true>, dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, ctx->stream, d, p);
    else hipLaunchKernelGGL(k_ntt_tile<false>, dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, ctx->stream, d, p);
    ZK_HIP(hipGetLastError());
}

__global__ void k_mul_rows(Fr* __restrict__ d, const Fr* __restrict__ f, size_t n, size_t total) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) d[i] = d[i] * f[i & (n - 1)];
}

// the tiles' dynamic LDS exceeds the default limit: raised once per device
static void ntt_tile_attributes(zk_ctx* ctx) {
    static PerDeviceOnce once;
    once.run(ctx->device, [] {
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
    });
}

// post: the factor applied with `scale` (null: 1 / 2^log_n) -- the blocks of a three-pass transform are scaled by 1 / n of the whole
static void ntt_core(zk_ctx* ctx, bool dit, Fr* d, unsigned log_n, bool inverse, bool scale, const Fr* d_pre, size_t batch, const Fr* post = nullptr, size_t pre_step = 0) {
    ntt_tile_attributes(ctx);
    auto tabs = ntt_get_tables(ctx, log_n);
    size_t n = (size_t)1 << log_n;
    const int32_t* tw = inverse ? tabs->tw29_inv.p : tabs->tw29_fwd.p;
    double pass_bytes = 64.0 * n * batch;  // read + write of every element
    if (!batch) return;
    const Fr post_f = post ? *post : tabs->n_inv;
    if (log_n > 2 * NTT_MAX_LOCAL_LOG) {
        // Three passes: n = 2^a1 x 2^22.  One more column pass of 2^a1-point transforms (stride 2^22; a tile = 2^a1 rows of
        // 2^(11 - a1) contiguous elements) with the twiddles w_n^(c brev(p)) on its store (DIF) / load (DIT), and the 2^a1
        // contiguous blocks of 2^22 as a batch of two-pass transforms.  Same orders as below: DIF natural -> bit-reversed, DIT back.
        const unsigned a1 = log_n - 2 * NTT_MAX_LOCAL_LOG, log_c = NTT_MAX_LOCAL_LOG - a1, log_blk = 2 * NTT_MAX_LOCAL_LOG;
        const Fr* mid = inverse ? tabs->mid_inv.p : tabs->mid_fwd.p;
        NttPass col{a1, log_c, log_blk - log_c, n, (size_t)1 << log_blk, (size_t)1 << log_c, tw, nullptr, nullptr, post_f, 0, 0};
        const size_t col_tiles = ((size_t)1 << (log_blk - log_c)) * batch;
        if (!dit) {
            col.mid = mid;
            launch_pass(ctx, false, d, col, col_tiles, "ntt_tile", pass_bytes);
            ntt_core(ctx, false, d, log_blk, inverse, scale, nullptr, batch << a1, &post_f);
        } else {
            if (d_pre) {   // (the two-pass form fuses this into its first load; at these sizes one more pass is 1 / 4 of the transform)
                hipLaunchKernelGGL(k_mul_rows, dim3(ceil_div(n * batch, 256)), dim3(256), 0, ctx->stream, d, d_pre, n, n * batch);
                ZK_HIP(hipGetLastError());
            }
            ntt_core(ctx, true, d, log_blk, inverse, false, nullptr, batch << a1);
            col.pre = mid;
            col.has_post = scale ? 1 : 0;
            launch_pass(ctx, true, d, col, col_tiles, "ntt_tile", pass_bytes);
        }
        return;
    }
    if (log_n <= NTT_MAX_LOCAL_LOG) {
        // batches of small transforms (the levels of interp.hip's tree): a tile takes as many whole transforms as fit its 2048 elements
        unsigned pack = 0;
        while ((!d_pre || pre_step) && log_n + pack < NTT_MAX_LOCAL_LOG && (batch >> (pack + 1)) << (pack + 1) == batch && (batch >> (pack + 1)) >= 256) ++pack;
        if (pack) {
            NttPass p{log_n, pack, 0, n << pack, 1, n << pack, tw, nullptr, d_pre, post_f, scale ? 1 : 0, 1, pre_step, n - 1};
            launch_pass(ctx, dit, d, p, batch >> pack, "ntt_tile", pass_bytes);
            return;
        }
        NttPass p{log_n, 0, 0, n, 1, n, tw, nullptr, d_pre, post_f, scale ? 1 : 0, 0, pre_step, n - 1};
        launch_pass(ctx, dit, d, p, batch, "ntt_tile", pass_bytes);
        return;
    }
    unsigned a = log_n - NTT_MAX_LOCAL_LOG;                 // column transform size 2^a
    unsigned log_c = NTT_MAX_LOCAL_LOG > a ? NTT_MAX_LOCAL_LOG - a : 0;  // columns per tile
    size_t r2 = (size_t)1 << NTT_MAX_LOCAL_LOG;
    const Fr* mid = inverse ? tabs->mid_inv.p : tabs->mid_fwd.p;
    NttPass col{a, log_c, NTT_MAX_LOCAL_LOG - log_c, n, r2, (size_t)1 << log_c, tw, nullptr, nullptr, post_f, 0, 0};
    NttPass row{NTT_MAX_LOCAL_LOG, 0, a, n, 1, r2, tw, nullptr, nullptr, post_f, 0, 0};
    size_t col_tiles = (r2 >> log_c) * batch, row_tiles = ((size_t)1 << a) * batch;
    if (!dit) {
        col.mid = mid;
        row.has_post = scale ? 1 : 0;
        if (d_pre) { col.pre = d_pre; col.pre_step = pre_step; col.pre_mask = n - 1; }   // periodic table on the first load (ntt_dif_pre)
        launch_pass(ctx, false, d, col, col_tiles, "ntt_tile", pass_bytes);
        launch_pass(ctx, false, d, row, row_tiles, "ntt_tile", pass_bytes);
    } else {
        row.pre = d_pre;
        row.mid = mid;
        col.has_post = scale ? 1 : 0;
        launch_pass(ctx, true, d, row, row_tiles, "ntt_tile", pass_bytes);
        launch_pass(ctx, true, d, col, col_tiles, "ntt_tile", pass_bytes);
    }
}

void ntt_dif(zk_ctx* ctx, Fr* d, unsigned log_n, bool inverse, bool scale, size_t batch) { ntt_core(ctx, false, d, log_n, inverse, scale, nullptr, batch); }

// The unscaled DIF transforms of the prove pipeline with their neighbouring element-wise kernels folded into the tile loads / stores
// (two-pass sizes, 2^12 .. 2^22 points; `batch` transforms whose first `half` belong to source / sink set 0, the rest to set 1):
//   * first pass: element = src_a[set][..] (x src_b[set][..] when given) instead of out[..] -- U.V on <w> and on the coset are formed in
//     the load, and the pass writes `out` while its sources stay intact (the SpMV outputs are needed twice);
//   * last pass: besides out[..] (Montgomery form, what the next transform reads) canonical(value x canon_k) goes to canon_out[set][..]
//     -- the scalars of the A and B inner products (k_scale_to_canonical's work).
// Saves two k_pointwise_mul, two k_scale_to_canonical and one 64 MB copy per proof (profiles/r5_experiments.txt item 10).
bool ntt_dif_fusable(unsigned log_n) { return log_n <= 2 * NTT_MAX_LOCAL_LOG; }
void ntt_dif_fused(zk_ctx* ctx, Fr* out, unsigned log_n, bool inverse, size_t batch, const NttFuse& f) {
    ZK_REQUIRE(ntt_dif_fusable(log_n) && batch >= 1 && f.half <= batch, ZK_ERR_SIZE, "ntt_dif_fused: at most two passes (2^22 points)");
    ntt_tile_attributes(ctx);
    auto tabs = ntt_get_tables(ctx, log_n);
    const size_t n = (size_t)1 << log_n, r2 = (size_t)1 << NTT_MAX_LOCAL_LOG;
    const int32_t* tw = inverse ? tabs->tw29_inv.p : tabs->tw29_fwd.p;
    if (log_n <= NTT_MAX_LOCAL_LOG) {   // one pass: a tile is one whole transform; load and store fusions meet in the same launch
        NttPass p{log_n, 0, 0, n, 1, n, tw, nullptr, nullptr, tabs->n_inv, 0, 0};
        p.fuse_half = (unsigned)f.half;
        for (int k = 0; k < 2; ++k) { p.src_a[k] = f.src_a[k]; p.src_b[k] = f.src_b[k]; p.canon_out[k] = f.canon_out[k]; }
        p.canon_k = f.canon_k.to_canonical();
        launch_pass(ctx, false, out, p, batch, "ntt_tile", 64.0 * n * batch);
        return;
    }
    const unsigned a = log_n - NTT_MAX_LOCAL_LOG, log_c = NTT_MAX_LOCAL_LOG > a ? NTT_MAX_LOCAL_LOG - a : 0;
    NttPass col{a, log_c, NTT_MAX_LOCAL_LOG - log_c, n, r2, (size_t)1 << log_c, tw, inverse ? tabs->mid_inv.p : tabs->mid_fwd.p, nullptr, tabs->n_inv, 0, 0};
    NttPass row{NTT_MAX_LOCAL_LOG, 0, a, n, 1, r2, tw, nullptr, nullptr, tabs->n_inv, 0, 0};
    col.fuse_half = row.fuse_half = (unsigned)f.half;
    for (int k = 0; k < 2; ++k) {
        col.src_a[k] = f.src_a[k]; col.src_b[k] = f.src_b[k];
        row.canon_out[k] = f.canon_out[k];
    }
    row.canon_k = f.canon_k.to_canonical();
    const double pass_bytes = 64.0 * n * batch;
    launch_pass(ctx, false, out, col, (r2 >> log_c) * batch, "ntt_tile", pass_bytes);
    launch_pass(ctx, false, out, row, ((size_t)1 << a) * batch, "ntt_tile", pass_bytes);
}
void ntt_dit(zk_ctx* ctx, Fr* d, unsigned log_n, bool inverse, bool scale, const Fr* d_pre, size_t batch) { ntt_core(ctx, true, d, log_n, inverse, scale, d_pre, batch); }
// forward DIF of `batch` transforms whose element i is first multiplied by table[i * step] (one table for the whole batch, fused into
// the first tile load); log_n <= 22
void ntt_dif_pre(zk_ctx* ctx, Fr* d, unsigned log_n, const Fr* table, size_t step, size_t batch) {
    ZK_REQUIRE(log_n <= 2 * NTT_MAX_LOCAL_LOG && step >= 1, ZK_ERR_SIZE, "ntt_dif_pre: at most 2^22 points");
    ntt_core(ctx, false, d, log_n, false, false, table, batch, nullptr, step);
}

__global__ void k_bitrev(const Fr* __restrict__ in, Fr* __restrict__ out, unsigned log_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ((size_t)1 << log_n)) out[brev((uint32_t)i, log_n)] = in[i];
}
void bitrev_permute(zk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_bitrev, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, out, log_n);
    ZK_HIP(hipGetLastError());
}

__global__ void k_pointwise_mul(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ out, size_t n) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
void fr_pointwise_mul(zk_ctx* ctx, const Fr* a, const Fr* b, Fr* out, size_t n) {
    if (!n) return;
    ProfScope ps(ctx, "fr_pointwise_mul", 96.0 * n);
    hipLaunchKernelGGL(k_pointwise_mul, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, a, b, out, n);
    ZK_HIP(hipGetLastError());
}

// ---- zk_lazy29_batch: the lazy radix-2^29 arithmetic on caller-supplied limb patterns (diagnostic) ----
// The multiplier of the bucket accumulation and of the NTT tiles (lazy29.cuh: FpR::mont / sqr / mont_diff / norm /
// store_exact; above: fr_reduce, fr_store_exact) normally only sees values the pipeline produces; the bounds it relies on
// are argued in comments.  This entry point feeds it the EXTREMES those comments allow (all limbs +-(2^29 - 1), 2^30 on one
// side, values next to k p) so that a column overflow shows up in a test instead of as a wrong proof on a rare input.
template <class PR>
__device__ __forceinline__ FpR<PR> lazy_get(const int32_t* p, size_t i) {
    FpR<PR> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.v[k] = p[i * 9 + k];
    return r;
}
template <class PR>
__global__ void k_lazy29(int op, const int32_t* __restrict__ a, const int32_t* __restrict__ b, const int32_t* __restrict__ c, const int32_t* __restrict__ d,
                         size_t n, Fp<PR>* __restrict__ out, int32_t* __restrict__ raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef FpR<PR> L;
    L x = lazy_get<PR>(a, i), r;
    if (op == ZK_LAZY_MONT) r = L::template mont<false>(x, lazy_get<PR>(b, i));
    else if (op == ZK_LAZY_SQR) r = x.sqr();
    else if (op == ZK_LAZY_MONT_DIFF) r = L::mont_diff(x, lazy_get<PR>(b, i), lazy_get<PR>(c, i), lazy_get<PR>(d, i));
    else if (op == ZK_LAZY_NORM) r = x.norm();
    else r = x;   // ZK_LAZY_STORE
    out[i] = r.store_exact();
    if (raw)
        for (int k = 0; k < 9; ++k) raw[i * 9 + k] = r.v[k];
}
__global__ void k_lazy29_fr(int op, const int32_t* __restrict__ a, size_t n, Fr* __restrict__ out, int32_t* __restrict__ raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FrL x = lazy_get<FrParams>(a, i), r = op == ZK_LAZY_FR_REDUCE ? fr_reduce(x) : x;
    out[i] = fr_store_exact(r);
    if (raw)
        for (int k = 0; k < 9; ++k) raw[i * 9 + k] = r.v[k];
}
void lazy29_batch(zk_ctx* ctx, int field, int op, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, size_t n, uint64_t* out, int32_t* raw_out) {
    ZK_REQUIRE(field == 0 || field == 1, ZK_ERR_ARG, "zk_lazy29_batch: field must be 0 (Fr) or 1 (Fq)");
    ZK_REQUIRE(op >= ZK_LAZY_MONT && op <= ZK_LAZY_FR_STORE, ZK_ERR_ARG, "zk_lazy29_batch: unknown op");
    ZK_REQUIRE(op < ZK_LAZY_FR_REDUCE || field == 0, ZK_ERR_ARG, "zk_lazy29_batch: fr_reduce / fr_store_exact exist for Fr only");
    const int operands = op == ZK_LAZY_MONT ? 2 : op == ZK_LAZY_MONT_DIFF ? 4 : 1;
    ZK_REQUIRE(out && (n == 0 || (a && (operands < 2 || b) && (operands < 4 || (c && d)))), ZK_ERR_ARG, "zk_lazy29_batch: null pointer");
    if (!n) return;
    const size_t limbs = n * 9;
    DevBuf<int32_t> da(limbs), db(operands >= 2 ? limbs : 0), dc(operands >= 4 ? limbs : 0), dd(operands >= 4 ? limbs : 0), draw(raw_out ? limbs : 0);
    DevBuf<Fr> dout(n);
    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemcpyAsync(da.p, a, limbs * 4, hipMemcpyHostToDevice, st));
    if (operands >= 2) ZK_HIP(hipMemcpyAsync(db.p, b, limbs * 4, hipMemcpyHostToDevice, st));
    if (operands >= 4) {
        ZK_HIP(hipMemcpyAsync(dc.p, c, limbs * 4, hipMemcpyHostToDevice, st));
        ZK_HIP(hipMemcpyAsync(dd.p, d, limbs * 4, hipMemcpyHostToDevice, st));
    }
    const dim3 grid(ceil_div(n, 256)), block(256);
    if (op >= ZK_LAZY_FR_REDUCE) hipLaunchKernelGGL(k_lazy29_fr, grid, block, 0, st, op, da.p, n, dout.p, draw.p);
    else if (field == 0) hipLaunchKernelGGL(k_lazy29<FrParams>, grid, block, 0, st, op, da.p, db.p, dc.p, dd.p, n, dout.p, draw.p);
    else hipLaunchKernelGGL(k_lazy29<FqParams>, grid, block, 0, st, op, da.p, db.p, dc.p, dd.p, n, reinterpret_cast<Fq*>(dout.p), draw.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(out, dout.p, n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    if (raw_out) ZK_HIP(hipMemcpyAsync(raw_out, draw.p, limbs * 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
}

// zk_ntt_fr: natural order in and out on a host buffer (== field::dft / idft semantics)
void ntt_host(zk_ctx* ctx, uint64_t* data, unsigned log_n, int inverse, int coset) {
    ZK_REQUIRE(data, ZK_ERR_ARG, "zk_ntt_fr: null data");
    ZK_REQUIRE(log_n <= NTT_MAX_LOG - (coset ? 1 : 0), ZK_ERR_SIZE, "zk_ntt_fr: log_n too large");
    size_t n = (size_t)1 << log_n;
    DevBuf<Fr> a(n), b(n), pw(coset ? n : 0);
    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    ZK_HIP(hipMemcpyAsync(a.p, data, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    fr_to_mont(ctx, a.p, a.p, n, flag.p);
    Fr g = coset ? host_root_of_unity(log_n + 1) : Fr::one();
    if (coset && !inverse) {
        fr_powers(ctx, g, Fr::one(), pw.p, n);
        fr_pointwise_mul(ctx, a.p, pw.p, a.p, n);
    }
    ntt_dif(ctx, a.p, log_n, inverse != 0, inverse != 0);
    bitrev_permute(ctx, a.p, b.p, log_n);
    if (coset && inverse) {
        fr_powers(ctx, g.inv(), Fr::one(), pw.p, n);
        fr_pointwise_mul(ctx, b.p, pw.p, b.p, n);
    }
    fr_from_mont(ctx, b.p, b.p, n);
    int hflag = 0;
    ZK_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!hflag, ZK_ERR_RANGE, "zk_ntt_fr: element >= r");
    ZK_HIP(hipMemcpy(data, b.p, n * sizeof(Fr), hipMemcpyDeviceToHost));
}

}  // namespace zk
