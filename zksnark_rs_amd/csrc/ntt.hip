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

std::shared_ptr<NttTables> ntt_get_tables(zk_ctx* ctx, unsigned log_n) {
    auto it = ctx->ntt_tables.find(log_n);
    if (it != ctx->ntt_tables.end()) return it->second;
    ZK_REQUIRE(log_n <= NTT_MAX_LOG, ZK_ERR_SIZE, "NTT size above 2^22 is not supported");
    auto t = std::make_shared<NttTables>();
    t->log_n = log_n;
    Fr w2048 = host_root_of_unity(NTT_MAX_LOCAL_LOG);
    t->tw_fwd.alloc(1024);
    t->tw_inv.alloc(1024);
    fr_powers(ctx, w2048, Fr::one(), t->tw_fwd.p, 1024);
    fr_powers(ctx, w2048.inv(), Fr::one(), t->tw_inv.p, 1024);
    if (log_n > NTT_MAX_LOCAL_LOG) {
        size_t n = (size_t)1 << log_n;
        unsigned a = log_n - NTT_MAX_LOCAL_LOG;
        Fr w = host_root_of_unity(log_n);
        t->mid_fwd.alloc(n);
        t->mid_inv.alloc(n);
        hipLaunchKernelGGL(k_mid_table, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, w, t->mid_fwd.p, a, NTT_MAX_LOCAL_LOG);
        hipLaunchKernelGGL(k_mid_table, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, w.inv(), t->mid_inv.p, a, NTT_MAX_LOCAL_LOG);
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
    hipLaunchKernelGGL(k_powers_brev, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, g, Fr::one(), t.coset_fwd_brev.p, t.log_n);
    hipLaunchKernelGGL(k_powers_brev, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, g.inv(), host_inv_pow2(1), t.coset_inv_brev_half.p, t.log_n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}

// ---- the tile kernel ---------------------------------------------------------------------
constexpr int TILE = 1 << NTT_MAX_LOCAL_LOG;  // elements per LDS tile
constexpr int NTT_THREADS = 256;

struct NttPass {
    unsigned log_rows, log_cols;
    size_t row_stride, tile_stride;
    const Fr* tw;    // w_2048^(+-k), k < 1024
    const Fr* mid;   // applied on store (data layout) or nullptr
    const Fr* pre;   // applied on load (data layout) or nullptr
    Fr post;
    int has_post;
};

__device__ __forceinline__ Fr lds_get(const uint32_t* lds, int e) {
    Fr r;
#pragma unroll
    for (int l = 0; l < 8; ++l) r.l[l] = lds[l * TILE + e];
    return r;
}
__device__ __forceinline__ void lds_put(uint32_t* lds, int e, const Fr& v) {
#pragma unroll
    for (int l = 0; l < 8; ++l) lds[l * TILE + e] = v.l[l];
}

template <bool DIT>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_tile(Fr* __restrict__ data, NttPass p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const unsigned log_rows = p.log_rows, log_cols = p.log_cols;
    const int rows = 1 << log_rows, cols = 1 << log_cols;
    const int elems = rows << log_cols;
    Fr* base = data + (size_t)blockIdx.x * p.tile_stride;
    const Fr* pre = p.pre ? p.pre + (size_t)blockIdx.x * p.tile_stride : nullptr;
    const Fr* mid = p.mid ? p.mid + (size_t)blockIdx.x * p.tile_stride : nullptr;

    // load: consecutive lanes walk the `cols` contiguous elements of a row, then the next row
    for (int idx = threadIdx.x; idx < elems; idx += NTT_THREADS) {
        int row = idx >> log_cols, col = idx & (cols - 1);
        size_t g = (size_t)row * p.row_stride + col;
        Fr v = base[g];
        if (pre) v = v * pre[g];
        lds_put(lds, (col << log_rows) + row, v);
    }
    __syncthreads();

    const int half_rows = rows >> 1;
    const unsigned tw_shift = NTT_MAX_LOCAL_LOG - log_rows;
    for (unsigned s = 0; s < log_rows; ++s) {
        for (int b = threadIdx.x; b < (elems >> 1); b += NTT_THREADS) {
            int col = b >> (log_rows - 1), t = b & (half_rows - 1);
            if (!DIT) {
                // decimation in frequency: (x, y) -> (x + y, (x - y) * w)
                unsigned lh = log_rows - 1 - s;  // log2(half)
                int k = t & ((1 << lh) - 1), grp = t >> lh;
                int i = (col << log_rows) + (grp << (lh + 1)) + k, j = i + (1 << lh);
                Fr w = p.tw[((size_t)k << s) << tw_shift];
                Fr x = lds_get(lds, i), y = lds_get(lds, j);
                lds_put(lds, i, x + y);
                lds_put(lds, j, (x - y) * w);
            } else {
                // decimation in time: (x, y) -> (x + w y, x - w y)
                int k = t & ((1 << s) - 1), grp = t >> s;
                int i = (col << log_rows) + (grp << (s + 1)) + k, j = i + (1 << s);
                Fr w = p.tw[((size_t)k << (log_rows - 1 - s)) << tw_shift];
                Fr x = lds_get(lds, i), y = lds_get(lds, j) * w;
                lds_put(lds, i, x + y);
                lds_put(lds, j, x - y);
            }
        }
        __syncthreads();
    }

    for (int idx = threadIdx.x; idx < elems; idx += NTT_THREADS) {
        int row = idx >> log_cols, col = idx & (cols - 1);
        size_t g = (size_t)row * p.row_stride + col;
        Fr v = lds_get(lds, (col << log_rows) + row);
        if (mid) v = v * mid[g];
        if (p.has_post) v = v * p.post;
        base[g] = v;
    }
}

static void launch_pass(zk_ctx* ctx, bool dit, Fr* d, const NttPass& p, size_t tiles, const char* name, double bytes) {
    ProfScope ps(ctx, name, bytes);
    size_t lds_bytes = (size_t)TILE * 32;
    if (dit) hipLaunchKernelGGL(k_ntt_tile<true>, dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, ctx->stream, d, p);
    else hipLaunchKernelGGL(k_ntt_tile<false>, dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, ctx->stream, d, p);
    ZK_HIP(hipGetLastError());
}

static void ntt_core(zk_ctx* ctx, bool dit, Fr* d, unsigned log_n, bool inverse, bool scale, const Fr* d_pre) {
    static bool attr_set = false;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE * 32));
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE * 32));
        attr_set = true;
    }
    auto tabs = ntt_get_tables(ctx, log_n);
    size_t n = (size_t)1 << log_n;
    const Fr* tw = inverse ? tabs->tw_inv.p : tabs->tw_fwd.p;
    double pass_bytes = 64.0 * n;  // read + write of every element
    if (log_n <= NTT_MAX_LOCAL_LOG) {
        NttPass p{log_n, 0, 1, n, tw, nullptr, d_pre, tabs->n_inv, scale ? 1 : 0};
        launch_pass(ctx, dit, d, p, 1, "ntt_tile", pass_bytes);
        return;
    }
    unsigned a = log_n - NTT_MAX_LOCAL_LOG;                 // column transform size 2^a
    unsigned log_c = NTT_MAX_LOCAL_LOG > a ? NTT_MAX_LOCAL_LOG - a : 0;  // columns per tile
    size_t r2 = (size_t)1 << NTT_MAX_LOCAL_LOG;
    const Fr* mid = inverse ? tabs->mid_inv.p : tabs->mid_fwd.p;
    NttPass col{a, log_c, r2, (size_t)1 << log_c, tw, nullptr, nullptr, tabs->n_inv, 0};
    NttPass row{NTT_MAX_LOCAL_LOG, 0, 1, r2, tw, nullptr, nullptr, tabs->n_inv, 0};
    size_t col_tiles = r2 >> log_c, row_tiles = (size_t)1 << a;
    if (!dit) {
        col.mid = mid;
        row.has_post = scale ? 1 : 0;
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

void ntt_dif(zk_ctx* ctx, Fr* d, unsigned log_n, bool inverse, bool scale) { ntt_core(ctx, false, d, log_n, inverse, scale, nullptr); }
void ntt_dit(zk_ctx* ctx, Fr* d, unsigned log_n, bool inverse, bool scale, const Fr* d_pre) { ntt_core(ctx, true, d, log_n, inverse, scale, d_pre); }

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
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
void fr_pointwise_mul(zk_ctx* ctx, const Fr* a, const Fr* b, Fr* out, size_t n) {
    if (!n) return;
    ProfScope ps(ctx, "fr_pointwise_mul", 96.0 * n);
    hipLaunchKernelGGL(k_pointwise_mul, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, a, b, out, n);
    ZK_HIP(hipGetLastError());
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
