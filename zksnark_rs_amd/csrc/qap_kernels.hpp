// qap_kernels.hpp -- launchers defined in qap.hip and used by the prove / setup pipelines.
#pragma once
#include "pipeline.hpp"

namespace zk {
void spmv(zk_ctx*, const DevCsr& m, const Fr* a, size_t a_len, Fr* out);
void dense_matvec(zk_ctx*, const Fr* M, const Fr* a, size_t rows, size_t n, Fr* out);
void h_combine(zk_ctx*, const Fr* x, const Fr* y, const Fr* tab, Fr half, Fr* out, size_t n);
void fr_scale_to_canonical(zk_ctx*, const Fr* in, Fr k, Fr* out, size_t n);
void fr_lincomb_to_canonical(zk_ctx*, const Fr* a, Fr ka, const Fr* b, Fr kb, Fr* out, size_t n);
void fr_sub_inplace(zk_ctx*, Fr* a, const Fr* b, size_t n);
void poly_divide(zk_ctx*, Fr* r, size_t len_r, const Fr* t, size_t d, const Fr* cinv, Fr* q);
void qap_ensure_tinv(zk_ctx*, zk_qap& q, size_t K, unsigned log_size);
void poly_divide_newton(zk_ctx*, const zk_qap& q, const Fr* r, size_t len_r, unsigned log_size, Fr* work, Fr* out);
}  // namespace zk
