// src/groth16/gpu.rs -- binding of libzkgpu.so inside the reference crate.  Not compiled in this
// repository: the image has no Rust toolchain; the same C ABI is exercised by tests/ through ctypes and through the
// C++ mirror include/zksnark.hpp (tests/cpp/reference_tests.cpp).
//! GPU prover behind the reference API: same argument order, borrow pattern and by-value
//! return as groth16::prove (mod.rs:213-217).
use super::{CoefficientPoly, Proof, QAP, SigmaG1, SigmaG2};
use super::fr::{FrLocal, G1Local, G2Local};
use std::os::raw::{c_int, c_uint, c_void};

#[repr(C)] pub struct ZkCtx { _p: [u8; 0] }
#[repr(C)] pub struct ZkCrs { _p: [u8; 0] }
#[repr(C)] pub struct ZkQap { _p: [u8; 0] }

#[repr(C)]
pub struct ZkCrsDesc {                                   // zk_crs_desc
    n: usize, m: usize, input: usize,
    alpha_g1: *const u64, beta_g1: *const u64, delta_g1: *const u64,
    xi_g1: *const u64, sum_gamma_g1: *const u64, sum_delta_g1: *const u64, xi_t_g1: *const u64,
    beta_g2: *const u64, gamma_g2: *const u64, delta_g2: *const u64, xi_g2: *const u64,
}

extern "C" {
    fn zk_ctx_create(device: c_int, out: *mut *mut ZkCtx) -> c_int;
    fn zk_ctx_destroy(ctx: *mut ZkCtx);
    fn zk_qap_upload_dense(ctx: *mut ZkCtx, u: *const u64, v: *const u64, w: *const u64, t: *const u64,
                           m: usize, n: usize, input: usize, out: *mut *mut ZkQap) -> c_int;
    fn zk_qap_free(q: *mut ZkQap);
    fn zk_crs_upload(ctx: *mut ZkCtx, desc: *const ZkCrsDesc, out: *mut *mut ZkCrs) -> c_int;
    fn zk_crs_free(c: *mut ZkCrs);
    fn zk_prove(ctx: *mut ZkCtx, crs: *const ZkCrs, qap: *const ZkQap, weights: *const u64, m: usize,
                r: *const u64, s: *const u64, proof_out: *mut u8) -> c_int;
    fn zk_last_error(ctx: *const ZkCtx) -> *const std::os::raw::c_char;
    // a stream of proofs: witnesses in page-locked host memory, two tickets in flight (zkgpu.h)
    fn zk_host_alloc(bytes: usize, out: *mut *mut std::os::raw::c_void) -> c_int;
    fn zk_host_free(p: *mut std::os::raw::c_void);
    fn zk_prove_submit_host(ctx: *mut ZkCtx, crs: *const ZkCrs, qap: *const ZkQap, weights: *const u64, m: usize,
                            r: *const u64, s: *const u64, ticket: *mut c_int) -> c_int;
    fn zk_prove_wait(ctx: *mut ZkCtx, ticket: c_int, proof_out: *mut u8) -> c_int;
}

fn check(ctx: *mut ZkCtx, rc: c_int) {
    // the reference panics on division by zero / zero divisor (fr.rs:54,69; field/mod.rs:440)
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(zk_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("{}", if rc == -5 { "Dividend must be non-zero".to_string() } else { msg });
    }
}

/// Device-resident copy of (QAP, CRS): upload once, prove many times.
pub struct GpuProver { ctx: *mut ZkCtx, qap: *mut ZkQap, crs: *mut ZkCrs, m: usize }

impl GpuProver {
    pub fn new(qap: &QAP<CoefficientPoly<FrLocal>>, sigma: (&SigmaG1<G1Local>, &SigmaG2<G2Local>)) -> Self {
        let (s1, s2) = sigma;
        let (m, n) = (qap.u.len(), qap.degree);
        // dense m x n coefficient matrices, zero padded (coefficient k of wire i at [i*n + k])
        let dense = |ps: &Vec<CoefficientPoly<FrLocal>>| -> Vec<u64> {
            let mut out = vec![0u64; m * n * 4];
            for (i, p) in ps.iter().enumerate() {
                for (k, c) in p.iter().take(n).enumerate() { out[(i * n + k) * 4..][..4].copy_from_slice(&fr_to_words(c)); }
            }
            out
        };
        let (u, v, w) = (dense(&qap.u), dense(&qap.v), dense(&qap.w));
        let t: Vec<u64> = qap.t.iter().flat_map(|c| fr_to_words(c).to_vec()).collect();
        let g1s = |ps: &[G1Local]| -> Vec<u64> { ps.iter().flat_map(|p| g1_to_words(p).to_vec()).collect() };
        let g2s = |ps: &[G2Local]| -> Vec<u64> { ps.iter().flat_map(|p| g2_to_words(p).to_vec()).collect() };
        let (xi1, sg, sd, xt, xi2) = (g1s(&s1.xi), g1s(&s1.sum_gamma), g1s(&s1.sum_delta), g1s(&s1.xi_t), g2s(&s2.xi));
        let (a1, b1, d1) = (g1_to_words(&s1.alpha), g1_to_words(&s1.beta), g1_to_words(&s1.delta));
        let (b2, g2, d2) = (g2_to_words(&s2.beta), g2_to_words(&s2.gamma), g2_to_words(&s2.delta));
        unsafe {
            let mut ctx = std::ptr::null_mut();
            assert_eq!(zk_ctx_create(0, &mut ctx), 0, "no MI355X visible (there is no CPU fallback)");
            let mut q = std::ptr::null_mut();
            check(ctx, zk_qap_upload_dense(ctx, u.as_ptr(), v.as_ptr(), w.as_ptr(), t.as_ptr(), m, n, qap.input, &mut q));
            let desc = ZkCrsDesc { n, m, input: qap.input,
                alpha_g1: a1.as_ptr(), beta_g1: b1.as_ptr(), delta_g1: d1.as_ptr(),
                xi_g1: xi1.as_ptr(), sum_gamma_g1: sg.as_ptr(), sum_delta_g1: sd.as_ptr(), xi_t_g1: xt.as_ptr(),
                beta_g2: b2.as_ptr(), gamma_g2: g2.as_ptr(), delta_g2: d2.as_ptr(), xi_g2: xi2.as_ptr() };
            let mut c = std::ptr::null_mut();
            check(ctx, zk_crs_upload(ctx, &desc, &mut c));
            GpuProver { ctx, qap: q, crs: c, m }
        }
    }

    /// groth16::prove with (r, s) injected (the reference draws them inside, mod.rs:231).
    pub fn prove_with_rs(&self, weights: &[FrLocal], r: FrLocal, s: FrLocal) -> Proof<G1Local, G2Local> {
        let w: Vec<u64> = weights.iter().flat_map(|c| fr_to_words(c).to_vec()).collect();
        let mut bytes = [0u8; 259];
        unsafe { check(self.ctx, zk_prove(self.ctx, self.crs, self.qap, w.as_ptr(), weights.len(),
                                          fr_to_words(&r).as_ptr(), fr_to_words(&s).as_ptr(), bytes.as_mut_ptr())); }
        Proof { a: g1_from_bytes(&bytes[0..65]), b: g2_from_bytes(&bytes[65..194]), c: g1_from_bytes(&bytes[194..259]) }
    }
}
impl GpuProver {
    /// Many proofs over one circuit: the 32 m-byte transfer of witness k+1 overlaps the inner products of proof k
    /// (zk_prove_submit_host / zk_prove_wait with two tickets in flight and page-locked staging buffers).
    pub fn prove_stream<'a, I>(&self, jobs: I) -> Vec<Proof<G1Local, G2Local>>
    where I: IntoIterator<Item = (&'a [FrLocal], FrLocal, FrLocal)> {
        let mut out = Vec::new();
        let mut inflight: std::collections::VecDeque<(c_int, usize)> = Default::default();
        let mut staging: [*mut std::os::raw::c_void; 2] = [std::ptr::null_mut(); 2];
        let mut cap = [0usize; 2];
        let mut finish = |t: c_int, out: &mut Vec<Proof<G1Local, G2Local>>| {
            let mut bytes = [0u8; 259];
            unsafe { check(self.ctx, zk_prove_wait(self.ctx, t, bytes.as_mut_ptr())); }
            out.push(Proof { a: g1_from_bytes(&bytes[0..65]), b: g2_from_bytes(&bytes[65..194]), c: g1_from_bytes(&bytes[194..259]) });
        };
        for (k, (weights, r, s)) in jobs.into_iter().enumerate() {
            if inflight.len() == 2 { let (t, _) = inflight.pop_front().unwrap(); finish(t, &mut out); }
            let slot = k % 2;            // the buffer of the proof waited for two submissions ago
            let need = weights.len() * 32;
            unsafe {
                if cap[slot] < need {
                    if !staging[slot].is_null() { zk_host_free(staging[slot]); }
                    assert_eq!(zk_host_alloc(need, &mut staging[slot]), 0);
                    cap[slot] = need;
                }
                let dst = std::slice::from_raw_parts_mut(staging[slot] as *mut u64, weights.len() * 4);
                for (i, c) in weights.iter().enumerate() { dst[4 * i..4 * i + 4].copy_from_slice(&fr_to_words(c)); }
                let mut t: c_int = -1;
                check(self.ctx, zk_prove_submit_host(self.ctx, self.crs, self.qap, staging[slot] as *const u64, weights.len(),
                                                     fr_to_words(&r).as_ptr(), fr_to_words(&s).as_ptr(), &mut t));
                inflight.push_back((t, slot));
            }
        }
        while let Some((t, _)) = inflight.pop_front() { finish(t, &mut out); }
        unsafe { for p in staging.iter() { if !p.is_null() { zk_host_free(*p); } } }
        out
    }
}
impl Drop for GpuProver {
    fn drop(&mut self) { unsafe { zk_crs_free(self.crs); zk_qap_free(self.qap); zk_ctx_destroy(self.ctx); } }
}

/// Drop-in for groth16::prove: identical signature (mod.rs:213-217).
pub fn prove(qap: &QAP<CoefficientPoly<FrLocal>>, sigma: (&SigmaG1<G1Local>, &SigmaG2<G2Local>), weights: &[FrLocal])
    -> Proof<G1Local, G2Local> {
    use super::Random;
    let (r, s) = (FrLocal::random_elem(), FrLocal::random_elem());
    GpuProver::new(qap, sigma).prove_with_rs(weights, r, s)
}
