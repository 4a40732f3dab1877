// interp.hpp -- interpolation through arbitrary nodes by a sub-product tree (interp.hip)
#pragma once
#include <memory>
#include <vector>
#include "kernels.hpp"

namespace zk {

#ifndef ZK_INTERP_BLOCK
#define ZK_INTERP_BLOCK 64
#endif
constexpr int INTERP_BLOCK = ZK_INTERP_BLOCK;   // leaves per flat bottom block

struct InterpTree {
    size_t n = 0;
    unsigned log_npad = 0;            // npad = 2^log_npad >= max(n, INTERP_BLOCK)
    DevBuf<Fr> roots, w;              // the nodes and 1 / N'(r_k), Montgomery
    DevBuf<Fr> qmat;                  // bottom blocks: [block][k][i] = coefficient i of N_block / (x - r_k), times w_k (node k outer, coefficient i inner:
                                      // k_interp_blocks, k_interp_scale_q, k_interp_bottom and gbasis.hip's k_gb_bottom all read this order; tests/test_arbitrary_roots.py
                                      // test_interp_single_block pins it through one block of known nodes)
    std::vector<DevBuf<Fr>> nev;      // level l: DIF images of the children's N (children of 64 << l leaves, padded to twice that): 2 npad each
    std::vector<DevBuf<Fr>> tws;      // level l: w_4s^j / 2s, j < 2s (s = 64 << l) -- the twist in front of the second half of a parent's doubled image,
                                      // carrying the 1 / 2s of the inverse transform before it (which therefore runs unscaled)
    DevBuf<Fr> t;                     // N_root = prod (x - r_k): n + 1 coefficients
};

// d_flag |= 16 when two roots coincide
std::shared_ptr<InterpTree> interp_build(zk_ctx*, const Fr* d_roots_mont, size_t n, int* d_flag);
// `count` vectors of values on the roots (vector v at d_values + v vstride, Montgomery) -> their interpolants' coefficients, vector v at
// d_out + v npad (npad entries, zero behind n); d_work: 3 count npad elements.  Everything on ctx->stream.
void interp_run(zk_ctx*, const InterpTree&, const Fr* d_values, size_t vstride, size_t count, Fr* d_work, Fr* d_out);
void interp_host(zk_ctx*, const uint64_t* roots, const uint64_t* values, size_t n, uint64_t* coeffs);

}  // namespace zk
