// src/groth16/gpu.rs -- binding of libzkgpu.so inside the reference crate.
//
// NOT compiled in this repository: the image has no Rust toolchain and the crate `bn 0.4.3` is not vendored.  What CAN
// be executed here is: every C-ABI call sequence below, in the same order and with the same argument conventions, by
// tests/cpp/rust_shim_sequence.c (run on the GPU by tests/test_cpp_api.py), and the byte-level conversions of module
// `bn_bytes` (big-endian coordinates <-> little-endian words, the Fq2 packing c1*q + c0), restated in C in the same file
// and checked against the public alt_bn128 vectors of tests/golden/alt_bn128.json.
//
// The module must live INSIDE the crate (src/groth16/gpu.rs, `pub mod gpu;` in src/groth16/mod.rs) because QAP, SigmaG1,
// SigmaG2 and Proof have private fields (mod.rs:60-67,105-128).  The newtypes of fr.rs wrap private bn values
// (fr.rs:8-16), so fr.rs additionally gets the six one-line accessors listed in INTEGRATION.md 2a:
//     impl FrLocal { pub(super) fn bn(&self) -> Fr { self.0 }  pub(super) fn from_bn(x: Fr) -> Self { FrLocal(x) } }
//     (the same pair for G1Local / G1 and G2Local / G2)
// and Cargo.toml gains `rustc-serialize = "0.3"` (already a dependency of bn; it is how bn exposes coordinates).
//
//! GPU Groth16 behind the reference API: `setup`, `prove`, `verify` with the argument order, borrow pattern and by-value
//! results of groth16::{setup, prove, verify} (mod.rs:134, 213-217, 299-303).
extern crate rustc_serialize;

use super::fr::{FrLocal, G1Local, G2Local};
use super::circuit::RootRepresentation;
use super::{CoefficientPoly, Proof, Random, SigmaG1, SigmaG2, QAP};
use std::cell::RefCell;
use std::collections::HashMap;
use std::os::raw::{c_char, c_int, c_uint, c_void};

// ------------------------------------------------------------------------------------------------
// C ABI (include/zkgpu.h)
// ------------------------------------------------------------------------------------------------
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
#[repr(C)]
pub struct ZkCrsOut {                                    // zk_crs_out (same field order, mutable; null = skip)
    alpha_g1: *mut u64, beta_g1: *mut u64, delta_g1: *mut u64,
    xi_g1: *mut u64, sum_gamma_g1: *mut u64, sum_delta_g1: *mut u64, xi_t_g1: *mut u64,
    beta_g2: *mut u64, gamma_g2: *mut u64, delta_g2: *mut u64, xi_g2: *mut u64,
}
#[repr(C)]
pub struct ZkSparseRows { ptr: *const u64, gate: *const u32, val: *const u64 }   // zk_sparse_rows: CSR by wire
#[repr(C)]
pub struct ZkQapSparseDesc { log_n: c_uint, m: usize, input: usize, u: ZkSparseRows, v: ZkSparseRows, w: ZkSparseRows }

extern "C" {
    fn zk_ctx_create(device: c_int, out: *mut *mut ZkCtx) -> c_int;
    fn zk_ctx_destroy(ctx: *mut ZkCtx);
    fn zk_last_error(ctx: *const ZkCtx) -> *const c_char;
    fn zk_qap_upload_dense(ctx: *mut ZkCtx, u: *const u64, v: *const u64, w: *const u64, t: *const u64,
                           m: usize, n: usize, input: usize, out: *mut *mut ZkQap) -> c_int;
    fn zk_qap_upload_sparse(ctx: *mut ZkCtx, desc: *const ZkQapSparseDesc, out: *mut *mut ZkQap) -> c_int;
    fn zk_qap_free(q: *mut ZkQap);
    fn zk_crs_upload(ctx: *mut ZkCtx, desc: *const ZkCrsDesc, out: *mut *mut ZkCrs) -> c_int;
    fn zk_crs_download(ctx: *mut ZkCtx, crs: *const ZkCrs, out: *const ZkCrsOut) -> c_int;
    fn zk_crs_free(c: *mut ZkCrs);
    fn zk_qap_upload_sparse_integers(ctx: *mut ZkCtx, desc: *const ZkQapSparseDesc, n: usize, out: *mut *mut ZkQap) -> c_int;
    fn zk_qap_upload_sparse_roots(ctx: *mut ZkCtx, desc: *const ZkQapSparseDesc, roots: *const u64, n: usize, out: *mut *mut ZkQap) -> c_int;
    fn zk_setup(ctx: *mut ZkCtx, qap: *const ZkQap, trapdoor: *const u64, out: *mut *mut ZkCrs) -> c_int;
    fn zk_prove(ctx: *mut ZkCtx, crs: *const ZkCrs, qap: *const ZkQap, weights: *const u64, m: usize,
                r: *const u64, s: *const u64, proof_out: *mut u8) -> c_int;
    fn zk_verify(ctx: *mut ZkCtx, crs: *const ZkCrs, inputs: *const u64, n_inputs: usize, proof: *const u8, ok: *mut c_int) -> c_int;
    // a stream of proofs: witnesses in page-locked host memory, two tickets in flight
    fn zk_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    fn zk_host_free(p: *mut c_void);
    fn zk_prove_submit_host(ctx: *mut ZkCtx, crs: *const ZkCrs, qap: *const ZkQap, weights: *const u64, m: usize,
                            r: *const u64, s: *const u64, ticket: *mut c_int) -> c_int;
    fn zk_prove_wait(ctx: *mut ZkCtx, ticket: c_int, proof_out: *mut u8) -> c_int;
    // several GPUs, one process (or thread) per GPU: a communicator + the scalar-exchange pipeline
    fn zk_device_count() -> c_int;
    fn zk_comm_unique_id(id_out: *mut u8) -> c_int;
    fn zk_comm_init(ctx: *mut ZkCtx, id: *const u8, rank: c_int, world: c_int, out: *mut *mut ZkComm) -> c_int;
    fn zk_comm_destroy(c: *mut ZkComm);
    fn zk_comm_set_timeout(c: *mut ZkComm, ms: std::os::raw::c_long) -> c_int;
    fn zk_comm_rccl_ranks(c: *const ZkComm) -> c_int;
    fn zk_mgpu_create(ctx: *mut ZkCtx, comm: *mut ZkComm, crs: *const ZkCrs, qap: *const ZkQap, out: *mut *mut ZkMgpu) -> c_int;
    fn zk_mgpu_push_host(p: *mut ZkMgpu, weights: *const u64, m: usize, r: *const u64, s: *const u64) -> c_int;
    fn zk_mgpu_pop(p: *mut ZkMgpu, proof_out: *mut u8) -> c_int;
    fn zk_mgpu_destroy(p: *mut ZkMgpu);
    fn zk_mgpu_last_error(p: *const ZkMgpu) -> *const c_char;
}
#[repr(C)] pub struct ZkComm { _p: [u8; 0] }
#[repr(C)] pub struct ZkMgpu { _p: [u8; 0] }

fn check(ctx: *mut ZkCtx, rc: c_int) {
    // the reference panics on division by zero / zero divisor (fr.rs:54,69; field/mod.rs:440); so does the shim
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(zk_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("{}", if rc == -5 { "Dividend must be non-zero".to_string() } else { msg });
    }
}

// ------------------------------------------------------------------------------------------------
// bn values <-> the ABI's element layout.  bn 0.4.3 keeps its field and group types opaque; what it does expose is
// rustc-serialize `Encodable` / `Decodable` (the README's own example serialises G1 through them):
//     Fr, Fq : 32 bytes, big-endian canonical integer
//     G1     : 0x00                      (identity)   |  0x04, x, y                 (affine, 32 + 32 bytes)
//     G2     : 0x00                      (identity)   |  0x04, X, Y                 (64 + 64 bytes), each Fq2 coordinate
//              packed as the 512-bit big-endian integer  c1 * q + c0
// [layout from the bn sources as remembered -- SURVEY F3: to be confirmed on a machine that has the crate; the unit
//  test at the end of this file does that in one `cargo test`].
// A byte sink / source implementing rustc_serialize::{Encoder, Decoder} captures exactly those bytes without bincode.
// ------------------------------------------------------------------------------------------------
mod bn_bytes {
    use super::rustc_serialize::{Decodable, Decoder, Encodable, Encoder};

    pub struct Sink(pub Vec<u8>);
    macro_rules! refuse { ($($name:ident : $t:ty),*) => { $(fn $name(&mut self, _: $t) -> Result<(), ()> { Err(()) })* } }
    impl Encoder for Sink {
        type Error = ();
        fn emit_nil(&mut self) -> Result<(), ()> { Ok(()) }
        fn emit_u8(&mut self, v: u8) -> Result<(), ()> { self.0.push(v); Ok(()) }
        refuse!(emit_usize: usize, emit_u64: u64, emit_u32: u32, emit_u16: u16, emit_isize: isize, emit_i64: i64, emit_i32: i32,
                emit_i16: i16, emit_i8: i8, emit_bool: bool, emit_f64: f64, emit_f32: f32, emit_char: char, emit_str: &str);
        fn emit_enum<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: &str, f: F) -> Result<(), ()> { f(self) }
        fn emit_enum_variant<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: &str, _: usize, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_enum_variant_arg<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_enum_struct_variant<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: &str, _: usize, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_enum_struct_variant_field<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: &str, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_struct<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: &str, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_struct_field<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: &str, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_tuple<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_tuple_arg<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_tuple_struct<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: &str, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_tuple_struct_arg<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_option<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, f: F) -> Result<(), ()> { f(self) }
        fn emit_option_none(&mut self) -> Result<(), ()> { Err(()) }
        fn emit_option_some<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, f: F) -> Result<(), ()> { f(self) }
        fn emit_seq<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_seq_elt<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_map<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_map_elt_key<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
        fn emit_map_elt_val<F: FnOnce(&mut Self) -> Result<(), ()>>(&mut self, _: usize, f: F) -> Result<(), ()> { f(self) }
    }

    pub struct Source<'a> { pub bytes: &'a [u8], pub pos: usize }
    macro_rules! refuse_read { ($($name:ident : $t:ty),*) => { $(fn $name(&mut self) -> Result<$t, String> { Err("not a byte".into()) })* } }
    impl<'a> Decoder for Source<'a> {
        type Error = String;
        fn read_nil(&mut self) -> Result<(), String> { Ok(()) }
        fn read_u8(&mut self) -> Result<u8, String> {
            let b = *self.bytes.get(self.pos).ok_or_else(|| "short input".to_string())?;
            self.pos += 1;
            Ok(b)
        }
        refuse_read!(read_usize: usize, read_u64: u64, read_u32: u32, read_u16: u16, read_isize: isize, read_i64: i64, read_i32: i32,
                     read_i16: i16, read_i8: i8, read_bool: bool, read_f64: f64, read_f32: f32, read_char: char, read_str: String);
        fn read_enum<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: &str, f: F) -> Result<T, String> { f(self) }
        fn read_enum_variant<T, F: FnMut(&mut Self, usize) -> Result<T, String>>(&mut self, _: &[&str], _: F) -> Result<T, String> { Err("enum".into()) }
        fn read_enum_variant_arg<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_enum_struct_variant<T, F: FnMut(&mut Self, usize) -> Result<T, String>>(&mut self, _: &[&str], _: F) -> Result<T, String> { Err("enum".into()) }
        fn read_enum_struct_variant_field<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: &str, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_struct<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: &str, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_struct_field<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: &str, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_tuple<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_tuple_arg<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_tuple_struct<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: &str, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_tuple_struct_arg<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_option<T, F: FnMut(&mut Self, bool) -> Result<T, String>>(&mut self, mut f: F) -> Result<T, String> { f(self, true) }
        fn read_seq<T, F: FnOnce(&mut Self, usize) -> Result<T, String>>(&mut self, _: F) -> Result<T, String> { Err("seq".into()) }
        fn read_seq_elt<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_map<T, F: FnOnce(&mut Self, usize) -> Result<T, String>>(&mut self, _: F) -> Result<T, String> { Err("map".into()) }
        fn read_map_elt_key<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: usize, f: F) -> Result<T, String> { f(self) }
        fn read_map_elt_val<T, F: FnOnce(&mut Self) -> Result<T, String>>(&mut self, _: usize, f: F) -> Result<T, String> { f(self) }
        fn error(&mut self, err: &str) -> String { err.to_string() }
    }

    pub fn encode<T: Encodable>(v: &T) -> Vec<u8> {
        let mut s = Sink(Vec::with_capacity(129));
        v.encode(&mut s).expect("bn value that does not encode to bytes");
        s.0
    }
    pub fn decode<T: Decodable>(bytes: &[u8]) -> T {
        let mut d = Source { bytes, pos: 0 };
        let v = T::decode(&mut d).expect("bytes that bn refuses (off-curve / out of range)");
        assert_eq!(d.pos, bytes.len(), "trailing bytes");
        v
    }

    // ---- 32-byte big-endian <-> [u64; 4] little-endian words (the ABI's Fr / Fq) ----
    pub fn be32_to_words(be: &[u8]) -> [u64; 4] {
        let mut w = [0u64; 4];
        for i in 0..4 { for b in 0..8 { w[i] = (w[i] << 8) | be[(3 - i) * 8 + b] as u64; } }
        w
    }
    pub fn words_to_be32(w: &[u64], out: &mut [u8]) {
        for i in 0..4 { for b in 0..8 { out[(3 - i) * 8 + b] = (w[i] >> (8 * (7 - b))) as u8; } }
    }

    // ---- bn's Fq2 packing: the 512-bit integer c1 * q + c0  <->  (c0, c1) ----
    pub const Q: [u64; 4] = [0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029];   // Fq modulus
    /// 64 bytes big-endian -> (c0, c1) by schoolbook long division of the 512-bit value by q (c1 < q because X < q^2)
    pub fn u512_to_fq2(be: &[u8]) -> ([u64; 4], [u64; 4]) {
        let mut rem = [0u64; 5];            // running remainder, < 2 q < 2^255: 4 words + guard
        let mut quo = [0u64; 4];
        for bit in 0..512 {
            let byte = be[bit / 8];
            let inb = ((byte >> (7 - bit % 8)) & 1) as u64;
            for k in (1..5).rev() { rem[k] = (rem[k] << 1) | (rem[k - 1] >> 63); }      // rem = rem * 2 + bit
            rem[0] = (rem[0] << 1) | inb;
            let ge = rem[4] != 0 || { let mut g = true; for k in (0..4).rev() { if rem[k] != Q[k] { g = rem[k] > Q[k]; break; } } g };
            for k in (1..4).rev() { quo[k] = (quo[k] << 1) | (quo[k - 1] >> 63); }
            quo[0] <<= 1;
            if ge {
                let mut borrow = 0u64;
                for k in 0..4 { let (d1, b1) = rem[k].overflowing_sub(Q[k]); let (d2, b2) = d1.overflowing_sub(borrow); rem[k] = d2; borrow = (b1 | b2) as u64; }
                rem[4] = rem[4].wrapping_sub(borrow);
                quo[0] |= 1;
            }
        }
        ([rem[0], rem[1], rem[2], rem[3]], quo)      // (c0 = X mod q, c1 = X div q)
    }
    /// (c0, c1) -> 64 bytes big-endian of c1 * q + c0
    pub fn fq2_to_u512(c0: &[u64], c1: &[u64], out: &mut [u8]) {
        let mut acc = [0u64; 8];
        for i in 0..4 {
            let mut carry = 0u128;
            for j in 0..4 {
                let t = acc[i + j] as u128 + (c1[i] as u128) * (Q[j] as u128) + carry;
                acc[i + j] = t as u64;
                carry = t >> 64;
            }
            acc[i + 4] = carry as u64;      // the row's top word is untouched so far
        }
        let mut carry = 0u128;
        for k in 0..8 { let t = acc[k] as u128 + if k < 4 { c0[k] as u128 } else { 0 } + carry; acc[k] = t as u64; carry = t >> 64; }
        for k in 0..8 { for b in 0..8 { out[(7 - k) * 8 + b] = (acc[k] >> (8 * (7 - b))) as u8; } }
    }
}

// ---- the five conversion helpers (INTEGRATION.md promised them; here they are) ----
pub fn fr_to_words(x: &FrLocal) -> [u64; 4] { bn_bytes::be32_to_words(&bn_bytes::encode(&x.bn())) }
pub fn fr_from_words(w: &[u64]) -> FrLocal {
    let mut be = [0u8; 32];
    bn_bytes::words_to_be32(w, &mut be);
    FrLocal::from_bn(bn_bytes::decode(&be))
}
/// G1 -> x | y as 2 x 4 words, identity = all zero (the ABI's convention)
pub fn g1_to_words(p: &G1Local) -> [u64; 8] {
    let b = bn_bytes::encode(&p.bn());
    let mut w = [0u64; 8];
    if b[0] == 4 {
        w[..4].copy_from_slice(&bn_bytes::be32_to_words(&b[1..33]));
        w[4..].copy_from_slice(&bn_bytes::be32_to_words(&b[33..65]));
    }
    w
}
/// G2 -> x.c0 | x.c1 | y.c0 | y.c1 as 4 x 4 words, identity = all zero
pub fn g2_to_words(p: &G2Local) -> [u64; 16] {
    let b = bn_bytes::encode(&p.bn());
    let mut w = [0u64; 16];
    if b[0] == 4 {
        let (x0, x1) = bn_bytes::u512_to_fq2(&b[1..65]);
        let (y0, y1) = bn_bytes::u512_to_fq2(&b[65..129]);
        w[0..4].copy_from_slice(&x0); w[4..8].copy_from_slice(&x1); w[8..12].copy_from_slice(&y0); w[12..16].copy_from_slice(&y1);
    }
    w
}
fn g1_from_words(w: &[u64]) -> G1Local {
    if w.iter().all(|&x| x == 0) { return G1Local::from_bn(bn_bytes::decode(&[0u8])); }
    let mut b = [0u8; 65];
    b[0] = 4;
    bn_bytes::words_to_be32(&w[0..4], &mut b[1..33]);
    bn_bytes::words_to_be32(&w[4..8], &mut b[33..65]);
    G1Local::from_bn(bn_bytes::decode(&b))
}
fn g2_from_words(w: &[u64]) -> G2Local {
    if w.iter().all(|&x| x == 0) { return G2Local::from_bn(bn_bytes::decode(&[0u8])); }
    let mut b = [0u8; 129];
    b[0] = 4;
    bn_bytes::fq2_to_u512(&w[0..4], &w[4..8], &mut b[1..65]);
    bn_bytes::fq2_to_u512(&w[8..12], &w[12..16], &mut b[65..129]);
    G2Local::from_bn(bn_bytes::decode(&b))
}
/// the 65-byte G1 block of a proof (0x04 | x | y, or 0x00 + zeros) -> G1Local.  Same bytes as bn's own encoding.
pub fn g1_from_bytes(p: &[u8]) -> G1Local {
    assert_eq!(p.len(), 65);
    if p[0] == 0 { G1Local::from_bn(bn_bytes::decode(&[0u8])) } else { G1Local::from_bn(bn_bytes::decode(p)) }
}
/// the 129-byte G2 block (0x04 | x.c1 | x.c0 | y.c1 | y.c0, EIP-197 order) -> G2Local
pub fn g2_from_bytes(p: &[u8]) -> G2Local {
    assert_eq!(p.len(), 129);
    if p[0] == 0 { return G2Local::from_bn(bn_bytes::decode(&[0u8])); }
    let mut w = [0u64; 16];
    w[4..8].copy_from_slice(&bn_bytes::be32_to_words(&p[1..33]));      // x.c1
    w[0..4].copy_from_slice(&bn_bytes::be32_to_words(&p[33..65]));     // x.c0
    w[12..16].copy_from_slice(&bn_bytes::be32_to_words(&p[65..97]));   // y.c1
    w[8..12].copy_from_slice(&bn_bytes::be32_to_words(&p[97..129]));   // y.c0
    g2_from_words(&w)
}
fn g1_to_bytes(p: &G1Local, out: &mut [u8]) {
    let w = g1_to_words(p);
    if w.iter().all(|&x| x == 0) { for b in out.iter_mut() { *b = 0; } return; }
    out[0] = 4;
    bn_bytes::words_to_be32(&w[0..4], &mut out[1..33]);
    bn_bytes::words_to_be32(&w[4..8], &mut out[33..65]);
}
fn g2_to_bytes(p: &G2Local, out: &mut [u8]) {
    let w = g2_to_words(p);
    if w.iter().all(|&x| x == 0) { for b in out.iter_mut() { *b = 0; } return; }
    out[0] = 4;
    bn_bytes::words_to_be32(&w[4..8], &mut out[1..33]);
    bn_bytes::words_to_be32(&w[0..4], &mut out[33..65]);
    bn_bytes::words_to_be32(&w[12..16], &mut out[65..97]);
    bn_bytes::words_to_be32(&w[8..12], &mut out[97..129]);
}
fn proof_from_bytes(b: &[u8; 259]) -> Proof<G1Local, G2Local> {
    Proof { a: g1_from_bytes(&b[0..65]), b: g2_from_bytes(&b[65..194]), c: g1_from_bytes(&b[194..259]) }
}
fn proof_to_bytes(p: &Proof<G1Local, G2Local>) -> [u8; 259] {
    let mut b = [0u8; 259];
    g1_to_bytes(&p.a, &mut b[0..65]);
    g2_to_bytes(&p.b, &mut b[65..194]);
    g1_to_bytes(&p.c, &mut b[194..259]);
    b
}

fn g1s(ps: &[G1Local]) -> Vec<u64> { ps.iter().flat_map(|p| g1_to_words(p).to_vec()).collect() }
fn g2s(ps: &[G2Local]) -> Vec<u64> { ps.iter().flat_map(|p| g2_to_words(p).to_vec()).collect() }
fn frs(xs: &[FrLocal]) -> Vec<u64> { xs.iter().flat_map(|x| fr_to_words(x).to_vec()).collect() }

// ------------------------------------------------------------------------------------------------
// Context, device-resident QAP / CRS
// ------------------------------------------------------------------------------------------------
struct Ctx(*mut ZkCtx);
impl Ctx {
    fn new() -> Ctx {
        let mut ctx = std::ptr::null_mut();
        let dev: c_int = std::env::var("ZKGPU_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        assert!(unsafe { zk_device_count() } > dev, "no MI355X visible (there is no CPU fallback)");
        assert_eq!(unsafe { zk_ctx_create(dev, &mut ctx) }, 0, "zk_ctx_create failed");
        Ctx(ctx)
    }
}
impl Drop for Ctx { fn drop(&mut self) { unsafe { zk_ctx_destroy(self.0) } } }

fn upload_dense(ctx: &Ctx, qap: &QAP<CoefficientPoly<FrLocal>>) -> *mut ZkQap {
    let (m, n) = (qap.u.len(), qap.degree);
    // dense m x n coefficient matrices, zero padded (coefficient k of wire i at [i*n + k]); CoefficientPoly derefs to [T]
    let dense = |ps: &Vec<CoefficientPoly<FrLocal>>| -> Vec<u64> {
        let mut out = vec![0u64; m * n * 4];
        for (i, p) in ps.iter().enumerate() {
            for (k, c) in p.iter().take(n).enumerate() { out[(i * n + k) * 4..][..4].copy_from_slice(&fr_to_words(c)); }
        }
        out
    };
    let (u, v, w) = (dense(&qap.u), dense(&qap.v), dense(&qap.w));
    let mut t = vec![0u64; (n + 1) * 4];
    for (k, c) in qap.t.iter().take(n + 1).enumerate() { t[4 * k..4 * k + 4].copy_from_slice(&fr_to_words(c)); }
    let mut q = std::ptr::null_mut();
    unsafe { check(ctx.0, zk_qap_upload_dense(ctx.0, u.as_ptr(), v.as_ptr(), w.as_ptr(), t.as_ptr(), m, n, qap.input, &mut q)); }
    q
}

fn upload_crs(ctx: &Ctx, s1: &SigmaG1<G1Local>, s2: &SigmaG2<G2Local>) -> (*mut ZkCrs, usize, usize, usize) {
    // dimensions as setup lays them out (mod.rs:146-194): |xi| = n, |sum_gamma| = l + 1, |sum_delta| = m - l - 1, |xi_t| = n - 1
    let (n, input) = (s1.xi.len(), s1.sum_gamma.len() - 1);
    let m = s1.sum_gamma.len() + s1.sum_delta.len();
    let (xi1, sg, sd, xt, xi2) = (g1s(&s1.xi), g1s(&s1.sum_gamma), g1s(&s1.sum_delta), g1s(&s1.xi_t), g2s(&s2.xi));
    let (a1, b1, d1) = (g1_to_words(&s1.alpha), g1_to_words(&s1.beta), g1_to_words(&s1.delta));
    let (b2, g2, d2) = (g2_to_words(&s2.beta), g2_to_words(&s2.gamma), g2_to_words(&s2.delta));
    let desc = ZkCrsDesc { n, m, input,
        alpha_g1: a1.as_ptr(), beta_g1: b1.as_ptr(), delta_g1: d1.as_ptr(),
        xi_g1: xi1.as_ptr(), sum_gamma_g1: sg.as_ptr(), sum_delta_g1: sd.as_ptr(), xi_t_g1: xt.as_ptr(),
        beta_g2: b2.as_ptr(), gamma_g2: g2.as_ptr(), delta_g2: d2.as_ptr(), xi_g2: xi2.as_ptr() };
    let mut c = std::ptr::null_mut();
    unsafe { check(ctx.0, zk_crs_upload(ctx.0, &desc, &mut c)); }   // range- and on-curve-checked on the GPU
    (c, n, m, input)
}

fn download_crs(ctx: &Ctx, crs: *const ZkCrs, n: usize, m: usize, l: usize) -> (SigmaG1<G1Local>, SigmaG2<G2Local>) {
    let (mut a1, mut b1, mut d1) = ([0u64; 8], [0u64; 8], [0u64; 8]);
    let (mut b2, mut g2, mut d2) = ([0u64; 16], [0u64; 16], [0u64; 16]);
    let (mut xi1, mut sg, mut sd, mut xt) = (vec![0u64; 8 * n], vec![0u64; 8 * (l + 1)], vec![0u64; 8 * (m - l - 1)], vec![0u64; 8 * (n - 1)]);
    let mut xi2 = vec![0u64; 16 * n];
    let out = ZkCrsOut { alpha_g1: a1.as_mut_ptr(), beta_g1: b1.as_mut_ptr(), delta_g1: d1.as_mut_ptr(),
        xi_g1: xi1.as_mut_ptr(), sum_gamma_g1: sg.as_mut_ptr(), sum_delta_g1: sd.as_mut_ptr(), xi_t_g1: xt.as_mut_ptr(),
        beta_g2: b2.as_mut_ptr(), gamma_g2: g2.as_mut_ptr(), delta_g2: d2.as_mut_ptr(), xi_g2: xi2.as_mut_ptr() };
    unsafe { check(ctx.0, zk_crs_download(ctx.0, crs, &out)); }
    let v1 = |w: &Vec<u64>| -> Vec<G1Local> { w.chunks(8).map(g1_from_words).collect() };
    (SigmaG1 { alpha: g1_from_words(&a1), beta: g1_from_words(&b1), delta: g1_from_words(&d1),
               xi: v1(&xi1), sum_gamma: v1(&sg), sum_delta: v1(&sd), xi_t: v1(&xt) },
     SigmaG2 { beta: g2_from_words(&b2), gamma: g2_from_words(&g2), delta: g2_from_words(&d2), xi: xi2.chunks(16).map(g2_from_words).collect() })
}

/// Device-resident copy of (QAP, CRS): upload once, prove many times.
pub struct GpuProver { ctx: Ctx, qap: *mut ZkQap, crs: *mut ZkCrs, n: usize, m: usize, input: usize }

impl GpuProver {
    /// from the reference's own types: dense QAP<CoefficientPoly<FrLocal>> + an existing CRS (any roots; up to 16384 gates)
    pub fn new(qap: &QAP<CoefficientPoly<FrLocal>>, sigma: (&SigmaG1<G1Local>, &SigmaG2<G2Local>)) -> Self {
        let ctx = Ctx::new();
        let q = upload_dense(&ctx, qap);
        let (crs, n, m, input) = upload_crs(&ctx, sigma.0, sigma.1);
        assert!(n == qap.degree && m == qap.u.len() && input == qap.input, "CRS and QAP dimensions differ");
        GpuProver { ctx, qap: q, crs, n, m, input }
    }

    /// Large circuits: the sparse form straight from a RootRepresentation (circuit/mod.rs:201-214) whose roots are
    /// 1, w, w^2, ... with w = 5^((r-1)/n), n = 2^log_n -- the dense QAP (3 m n field elements) is never built -- and a
    /// CRS made on the GPU (groth16::setup, mod.rs:134-197).  Returns the prover and the CRS in the reference's types.
    pub fn from_root_rep<R: RootRepresentation<FrLocal>>(rr: &R, log_n: u32) -> (Self, (SigmaG1<G1Local>, SigmaG2<G2Local>)) {
        let n = 1usize << log_n;
        let mut index: HashMap<[u64; 4], u32> = HashMap::with_capacity(n);
        // w = 5^((r-1)/n): built from the published 2^28-th root of unity by repeated squaring
        let mut w = "19103219067921713944291392827692070036145651957329286315305642004821462161904".parse::<FrLocal>().ok().expect("omega");
        for _ in log_n..28 { w = w * w; }
        let mut p = FrLocal::from(1usize);
        for (j, root) in rr.roots().enumerate() {
            assert!(j < n && root == p, "zk_qap_upload_sparse needs the roots 1, w, w^2, ... (use GpuProver::new for other roots)");
            index.insert(fr_to_words(&root), j as u32);
            p = p * w;
        }
        assert_eq!(index.len(), n, "the number of gates must be 2^log_n");
        Self::from_rows(rr, &index, n, Some(log_n), None, None)
    }

    /// The circuits the reference itself produces: a RootRepresentation over the roots 1, 2, .., n (ASTParser, circuit/mod.rs:517),
    /// any n up to 2^23, with the reference's proof bytes (zk_qap_upload_sparse_integers: nothing is interpolated).  Builder
    /// circuits (CircuitInstance, keccak) arrive here too once `From<&CircuitInstance> for DummyRep` stops pre-filling u, v, w with
    /// num_wires empty rows before pushing the real ones (circuit/mod.rs:163-165 then :186-188 -- SURVEY F8: as shipped, the
    /// first num_wires polynomials are zero and every proof verifies vacuously; `Vec::with_capacity` is the fix) and is
    /// instantiated with sub_circuit_point = |id| FrLocal::from(id + 1).
    pub fn from_root_rep_integers<R: RootRepresentation<FrLocal>>(rr: &R) -> (Self, (SigmaG1<G1Local>, SigmaG2<G2Local>)) {
        let mut index: HashMap<[u64; 4], u32> = HashMap::new();
        let mut k = FrLocal::from(1usize);
        let mut n = 0usize;
        for root in rr.roots() {
            assert!(root == k, "zk_qap_upload_sparse_integers needs the roots 1, 2, 3, ...");
            index.insert(fr_to_words(&root), n as u32);
            k = k + FrLocal::from(1usize);
            n += 1;
        }
        Self::from_rows(rr, &index, n, None, None, None)
    }

    /// ANY RootRepresentation (circuit/mod.rs:201-214: `roots()` is caller data, e.g. DummyRep's, dummy_rep.rs:47): the rows over the
    /// caller's distinct roots, at any size up to 2^22 gates, with the reference's proof bytes (zk_qap_upload_sparse_roots: the prover
    /// interpolates U and V per proof by a sub-product tree, nothing is interpolated per wire).  `sigma`: a CRS
    /// the reference's own `setup` made for the circuit, or None to make one on the GPU.
    pub fn from_root_rep_any<R: RootRepresentation<FrLocal>>(rr: &R, sigma: Option<(&SigmaG1<G1Local>, &SigmaG2<G2Local>)>)
        -> (Self, (SigmaG1<G1Local>, SigmaG2<G2Local>)) {
        let mut index: HashMap<[u64; 4], u32> = HashMap::new();
        let mut roots: Vec<u64> = Vec::new();
        for (j, root) in rr.roots().enumerate() {
            let w = fr_to_words(&root);
            assert!(index.insert(w, j as u32).is_none(), "the roots of a RootRepresentation must be distinct");
            roots.extend_from_slice(&w);
        }
        let n = index.len();
        Self::from_rows(rr, &index, n, None, sigma, Some(&roots))
    }

    /// The same circuits with a CRS the reference's own `setup` made (groth16::prove takes any (&SigmaG1, &SigmaG2), mod.rs:213-217):
    /// the library derives the Lagrange-basis points it multiplies with from [x^i]_1, [x^i]_2, [x^i t(x)/delta]_1 at the first proof
    /// (public linear combinations; once per CRS, O(n^2): n <= 2^16 + 2^10 gates); above that size it proves the same bytes
    /// through the sub-product tree of the roots 1..n.
    pub fn with_sigma_integers<R: RootRepresentation<FrLocal>>(rr: &R, sigma: (&SigmaG1<G1Local>, &SigmaG2<G2Local>)) -> Self {
        let mut index: HashMap<[u64; 4], u32> = HashMap::new();
        let mut k = FrLocal::from(1usize);
        let mut n = 0usize;
        for root in rr.roots() {
            assert!(root == k, "zk_qap_upload_sparse_integers needs the roots 1, 2, 3, ...");
            index.insert(fr_to_words(&root), n as u32);
            k = k + FrLocal::from(1usize);
            n += 1;
        }
        Self::from_rows(rr, &index, n, None, Some(sigma), None).0
    }

    fn from_rows<R: RootRepresentation<FrLocal>>(rr: &R, index: &HashMap<[u64; 4], u32>, n: usize, log_n: Option<u32>,
                                                 given: Option<(&SigmaG1<G1Local>, &SigmaG2<G2Local>)>, roots: Option<&Vec<u64>>)
        -> (Self, (SigmaG1<G1Local>, SigmaG2<G2Local>)) {
        struct Rows { ptr: Vec<u64>, gate: Vec<u32>, val: Vec<u64> }
        let collect = |rows: R::Row| -> Rows {
            let mut r = Rows { ptr: vec![0], gate: Vec::new(), val: Vec::new() };
            for col in rows {
                for (root, value) in col {
                    r.gate.push(index[&fr_to_words(&root)]);
                    r.val.extend_from_slice(&fr_to_words(&value));
                }
                r.ptr.push(r.gate.len() as u64);
            }
            r
        };
        let (u, v, wr) = (collect(rr.u()), collect(rr.v()), collect(rr.w()));
        let m = u.ptr.len() - 1;
        assert!(v.ptr.len() - 1 == m && wr.ptr.len() - 1 == m);      // fr.rs:157-158
        let raw = |r: &Rows| ZkSparseRows { ptr: r.ptr.as_ptr(), gate: r.gate.as_ptr(), val: r.val.as_ptr() };
        let desc = ZkQapSparseDesc { log_n: log_n.unwrap_or(0) as c_uint, m, input: rr.input(), u: raw(&u), v: raw(&v), w: raw(&wr) };
        let ctx = Ctx::new();
        let (mut q, mut crs) = (std::ptr::null_mut(), std::ptr::null_mut());
        let td: Vec<u64> = (0..5).flat_map(|_| fr_to_words(&FrLocal::random_elem()).to_vec()).collect();   // alpha, beta, gamma, delta, x (mod.rs:139-145)
        unsafe {
            match (log_n, roots) {
                (Some(_), _) => check(ctx.0, zk_qap_upload_sparse(ctx.0, &desc, &mut q)),
                (None, Some(r)) => check(ctx.0, zk_qap_upload_sparse_roots(ctx.0, &desc, r.as_ptr(), n, &mut q)),
                (None, None) => check(ctx.0, zk_qap_upload_sparse_integers(ctx.0, &desc, n, &mut q)),
            }
            match given {
                Some((s1, s2)) => {
                    let (c, cn, cm, cl) = upload_crs(&ctx, s1, s2);
                    assert!(cn == n && cm == m && cl == rr.input(), "CRS and circuit dimensions differ");
                    crs = c;
                }
                None => check(ctx.0, zk_setup(ctx.0, q, td.as_ptr(), &mut crs)),
            }
        }
        let sigma = download_crs(&ctx, crs, n, m, rr.input());
        (GpuProver { ctx, qap: q, crs, n, m, input: rr.input() }, sigma)
    }

    /// groth16::prove with (r, s) injected (the reference draws them inside, mod.rs:231).
    pub fn prove_with_rs(&self, weights: &[FrLocal], r: FrLocal, s: FrLocal) -> Proof<G1Local, G2Local> {
        let w = frs(weights);
        let mut bytes = [0u8; 259];
        unsafe { check(self.ctx.0, zk_prove(self.ctx.0, self.crs, self.qap, w.as_ptr(), weights.len(),
                                            fr_to_words(&r).as_ptr(), fr_to_words(&s).as_ptr(), bytes.as_mut_ptr())); }
        proof_from_bytes(&bytes)
    }
    pub fn prove(&self, weights: &[FrLocal]) -> Proof<G1Local, G2Local> {
        self.prove_with_rs(weights, FrLocal::random_elem(), FrLocal::random_elem())
    }
    /// groth16::verify against the device CRS
    pub fn verify(&self, inputs: &[FrLocal], proof: &Proof<G1Local, G2Local>) -> bool {
        let (x, bytes, mut ok) = (frs(inputs), proof_to_bytes(proof), 0 as c_int);
        unsafe { check(self.ctx.0, zk_verify(self.ctx.0, self.crs, x.as_ptr(), inputs.len(), bytes.as_ptr(), &mut ok)); }
        ok == 1
    }
    /// Many proofs over one circuit: the 32 m-byte transfer of witness k+1 overlaps the inner products of proof k
    /// (zk_prove_submit_host / zk_prove_wait with two tickets in flight and page-locked staging buffers).
    pub fn prove_stream<'a, I>(&self, jobs: I) -> Vec<Proof<G1Local, G2Local>>
    where I: IntoIterator<Item = (&'a [FrLocal], FrLocal, FrLocal)> {
        let mut out = Vec::new();
        let mut inflight: std::collections::VecDeque<c_int> = Default::default();
        let mut staging: [*mut c_void; 2] = [std::ptr::null_mut(); 2];
        let mut cap = [0usize; 2];
        let finish = |t: c_int, out: &mut Vec<Proof<G1Local, G2Local>>| {
            let mut bytes = [0u8; 259];
            unsafe { check(self.ctx.0, zk_prove_wait(self.ctx.0, t, bytes.as_mut_ptr())); }
            out.push(proof_from_bytes(&bytes));
        };
        for (k, (weights, r, s)) in jobs.into_iter().enumerate() {
            if inflight.len() == 2 { let t = inflight.pop_front().unwrap(); finish(t, &mut out); }
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
                check(self.ctx.0, zk_prove_submit_host(self.ctx.0, self.crs, self.qap, staging[slot] as *const u64, weights.len(),
                                                       fr_to_words(&r).as_ptr(), fr_to_words(&s).as_ptr(), &mut t));
                inflight.push_back(t);
            }
        }
        while let Some(t) = inflight.pop_front() { finish(t, &mut out); }
        unsafe { for p in staging.iter() { if !p.is_null() { zk_host_free(*p); } } }
        out
    }
    pub fn dims(&self) -> (usize, usize, usize) { (self.n, self.m, self.input) }
}
impl Drop for GpuProver {
    fn drop(&mut self) { unsafe { zk_crs_free(self.crs); zk_qap_free(self.qap); } }   // self.ctx drops afterwards (field order)
}

// ------------------------------------------------------------------------------------------------
// prove() over all GPUs of a node.  One MultiGpuProver per GPU (a thread or a process each); rank 0 draws the 128-byte id
// with MultiGpuProver::unique_id() and hands it to the others by any channel.  A ROUND is `world` proofs, one per rank:
// every rank pushes its own witness, the library exchanges the scalars of the four inner products so that rank g multiplies
// its own 1/world of the CRS points for every proof of the round (RCCL over xGMI, linked into libzkgpu.so), returns the
// 768-byte partial sums to the proofs' owners and assembles.  Every rank must make the same sequence of calls.
// ------------------------------------------------------------------------------------------------
pub struct MultiGpuProver { inner: GpuProver, comm: *mut ZkComm, mgpu: *mut ZkMgpu, world: usize }

impl MultiGpuProver {
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        assert_eq!(unsafe { zk_comm_unique_id(id.as_mut_ptr()) }, 0, "zk_comm_unique_id failed");
        id
    }
    /// `prover`: this rank's device copy of (QAP, CRS) in the roots-of-unity form (GpuProver::from_root_rep, made on the device
    /// `ZKGPU_DEVICE` selects); collective over the `world` ranks.
    pub fn new(prover: GpuProver, id: &[u8; 128], rank: usize, world: usize) -> Self {
        let (mut comm, mut mgpu) = (std::ptr::null_mut(), std::ptr::null_mut());
        unsafe {
            check(prover.ctx.0, zk_comm_init(prover.ctx.0, id.as_ptr(), rank as c_int, world as c_int, &mut comm));
            check(prover.ctx.0, zk_mgpu_create(prover.ctx.0, comm, prover.crs, prover.qap, &mut mgpu));
        }
        MultiGpuProver { inner: prover, comm, mgpu, world }
    }
    fn check(&self, rc: c_int) {
        if rc != 0 {
            panic!("{}", unsafe { std::ffi::CStr::from_ptr(zk_mgpu_last_error(self.mgpu)) }.to_string_lossy());
        }
    }
    /// This rank's share of a stream of rounds: job k is the proof this rank owns in round k.  Witnesses are staged in
    /// page-locked memory, two rounds are pushed ahead of every pop (the schedule the throughput numbers are quoted on).
    pub fn prove_stream<'a, I>(&self, jobs: I) -> Vec<Proof<G1Local, G2Local>>
    where I: IntoIterator<Item = (&'a [FrLocal], FrLocal, FrLocal)> {
        let mut out = Vec::new();
        let mut staging: [*mut c_void; 4] = [std::ptr::null_mut(); 4];   // a round's witness is read before its pop; <= 3 in flight
        let mut cap = [0usize; 4];
        let mut in_flight = 0usize;
        let pop = |out: &mut Vec<Proof<G1Local, G2Local>>| {
            let mut bytes = [0u8; 259];
            self.check(unsafe { zk_mgpu_pop(self.mgpu, bytes.as_mut_ptr()) });
            out.push(proof_from_bytes(&bytes));
        };
        for (k, (weights, r, s)) in jobs.into_iter().enumerate() {
            if in_flight == 3 { pop(&mut out); in_flight -= 1; }
            let slot = k % 4;
            let need = weights.len() * 32;
            unsafe {
                if cap[slot] < need {
                    if !staging[slot].is_null() { zk_host_free(staging[slot]); }
                    assert_eq!(zk_host_alloc(need, &mut staging[slot]), 0);
                    cap[slot] = need;
                }
                let dst = std::slice::from_raw_parts_mut(staging[slot] as *mut u64, weights.len() * 4);
                for (i, c) in weights.iter().enumerate() { dst[4 * i..4 * i + 4].copy_from_slice(&fr_to_words(c)); }
                self.check(zk_mgpu_push_host(self.mgpu, staging[slot] as *const u64, weights.len(),
                                             fr_to_words(&r).as_ptr(), fr_to_words(&s).as_ptr()));
            }
            in_flight += 1;
        }
        while in_flight > 0 { pop(&mut out); in_flight -= 1; }
        unsafe { for p in staging.iter() { if !p.is_null() { zk_host_free(*p); } } }
        out
    }
    /// One proof per rank and call: the reference's prove(), `world` of them at a time.
    pub fn prove(&self, weights: &[FrLocal]) -> Proof<G1Local, G2Local> {
        self.prove_stream(std::iter::once((weights, FrLocal::random_elem(), FrLocal::random_elem()))).pop().unwrap()
    }
    pub fn world(&self) -> usize { self.world }
    /// Bound of every wait for a peer (default 120 s): a pop whose round does not complete in time panics with the library's
    /// message instead of blocking for ever, and the communicator is aborted.
    pub fn set_timeout_ms(&self, ms: u64) { unsafe { zk_comm_set_timeout(self.comm, ms as std::os::raw::c_long); } }
    /// Ranks of the RCCL communicator as RCCL counts them (0 when `world` is 1).
    pub fn rccl_ranks(&self) -> usize { unsafe { zk_comm_rccl_ranks(self.comm) as usize } }
    pub fn single(&self) -> &GpuProver { &self.inner }
}
impl Drop for MultiGpuProver {
    fn drop(&mut self) { unsafe { zk_mgpu_destroy(self.mgpu); zk_comm_destroy(self.comm); } }   // self.inner (ctx, crs, qap) drops afterwards
}

// ------------------------------------------------------------------------------------------------
// The reference's three functions
// ------------------------------------------------------------------------------------------------
/// groth16::setup (mod.rs:134): the trapdoor is drawn here (thread_rng, as mod.rs:139-145), every group element is computed
/// on the GPU and handed back in the reference's types.
pub fn setup(qap: &QAP<CoefficientPoly<FrLocal>>) -> (SigmaG1<G1Local>, SigmaG2<G2Local>) {
    let ctx = Ctx::new();
    let q = upload_dense(&ctx, qap);
    let td: Vec<u64> = (0..5).flat_map(|_| fr_to_words(&FrLocal::random_elem()).to_vec()).collect();
    let mut crs = std::ptr::null_mut();
    unsafe { check(ctx.0, zk_setup(ctx.0, q, td.as_ptr(), &mut crs)); }
    let sigma = download_crs(&ctx, crs, qap.degree, qap.u.len(), qap.input);
    unsafe { zk_crs_free(crs); zk_qap_free(q); }
    sigma
}

thread_local! {
    // the prover of the most recent (qap, sigma) pair of this thread: groth16::prove takes both by reference on every call
    // (mod.rs:213-217), and re-uploading 3 m n coefficients + the CRS per proof would dominate.  Keyed by the addresses
    // AND a fingerprint of the contents, so a freed-and-reallocated object at the same address is not mistaken for the old one.
    static CACHED: RefCell<Option<((usize, usize, usize, [u64; 12]), GpuProver)>> = RefCell::new(None);
}
fn fingerprint(qap: &QAP<CoefficientPoly<FrLocal>>, s1: &SigmaG1<G1Local>) -> [u64; 12] {
    let mut f = [0u64; 12];
    f[0] = qap.u.len() as u64; f[1] = qap.degree as u64; f[2] = qap.input as u64; f[3] = s1.xi.len() as u64;
    if let Some(c) = qap.t.iter().nth(qap.degree / 2) { f[4..8].copy_from_slice(&fr_to_words(c)); }
    f[8..12].copy_from_slice(&g1_to_words(&s1.delta)[0..4]);
    f
}

/// Drop-in for groth16::prove: identical signature (mod.rs:213-217).  The device copies of the QAP and the CRS are kept
/// between calls with the same arguments.
pub fn prove(qap: &QAP<CoefficientPoly<FrLocal>>, sigma: (&SigmaG1<G1Local>, &SigmaG2<G2Local>), weights: &[FrLocal])
    -> Proof<G1Local, G2Local> {
    let key = (qap as *const _ as usize, sigma.0 as *const _ as usize, sigma.1 as *const _ as usize, fingerprint(qap, sigma.0));
    CACHED.with(|c| {
        let mut slot = c.borrow_mut();
        let hit = match *slot { Some((ref k, _)) => *k == key, None => false };
        if !hit { *slot = Some((key, GpuProver::new(qap, sigma))); }
        slot.as_ref().unwrap().1.prove(weights)
    })
}

/// groth16::verify (mod.rs:299-303; `P` is the reference's phantom parameter, call sites write verify::<CoefficientPoly<FrLocal>, ..>)
pub fn verify<P>(sigma: (&SigmaG1<G1Local>, &SigmaG2<G2Local>), inputs: &[FrLocal], proof: &Proof<G1Local, G2Local>) -> bool {
    let ctx = Ctx::new();
    let (crs, _, _, _) = upload_crs(&ctx, sigma.0, sigma.1);
    let (x, bytes, mut ok) = (frs(inputs), proof_to_bytes(proof), 0 as c_int);
    unsafe {
        check(ctx.0, zk_verify(ctx.0, crs, x.as_ptr(), inputs.len(), bytes.as_ptr(), &mut ok));
        zk_crs_free(crs);
    }
    ok == 1
}

// ------------------------------------------------------------------------------------------------
// One `cargo test` on a machine with the crate settles every [recollection] above
// ------------------------------------------------------------------------------------------------
#[cfg(test)]
mod tests {
    use super::*;
    use super::super::fr::*;
    use super::super::EllipticEncryptable;

    #[test]
    fn bn_byte_layout_is_what_the_shim_assumes() {
        // Fr: 32 bytes big-endian
        let seven = FrLocal::from(7usize);
        assert_eq!(fr_to_words(&seven), [7, 0, 0, 0]);
        assert!(fr_from_words(&[7, 0, 0, 0]) == seven);
        // G1: 69 * (1, 2) (fr.rs:106-109) round-trips, and 2 * (1, 2) is the public EIP-196 value (tests/golden/alt_bn128.json, cdetrio11)
        let g = FrLocal::from(1usize).encrypt_g1();
        assert!(g1_from_words(&g1_to_words(&g)) == g);
        let two_g = g1_from_words(&[1, 0, 0, 0, 2, 0, 0, 0]) + g1_from_words(&[1, 0, 0, 0, 2, 0, 0, 0]);
        assert_eq!(g1_to_words(&two_g)[0], 0xd3c208c16d87cfd3);
        // G2: packing c1 * q + c0 round-trips through bn's decoder (which checks the curve equation)
        let h = FrLocal::from(1usize).encrypt_g2();
        assert!(g2_from_words(&g2_to_words(&h)) == h);
        // identity
        assert!(g1_to_words(&(g - g)).iter().all(|&x| x == 0));
    }

    #[test]
    fn simple_circuit_on_the_gpu() {
        // lib.rs:156-190 with gpu::{setup, prove, verify} in place of the generic functions
        use super::super::circuit::{ASTParser, TryParse};
        let code = include_str!("../../test_programs/simple.zk");
        let qap: QAP<CoefficientPoly<FrLocal>> = ASTParser::try_parse(code).unwrap().into();
        let weights = super::super::circuit::weights(code, &[FrLocal::from(3usize), FrLocal::from(2usize), FrLocal::from(4usize)]).unwrap();
        let (s1, s2) = setup(&qap);
        let proof = prove(&qap, (&s1, &s2), &weights);
        assert!(verify::<CoefficientPoly<FrLocal>>((&s1, &s2), &[FrLocal::from(2usize), FrLocal::from(34usize)], &proof));
        assert!(!verify::<CoefficientPoly<FrLocal>>((&s1, &s2), &[FrLocal::from(2usize), FrLocal::from(25usize)], &proof));
        // and interchangeably with the CPU functions: a GPU proof verifies on the CPU path, a CPU proof on the GPU path
        assert!(super::super::verify::<CoefficientPoly<FrLocal>, _, _, _, _>((s1_ref(&s1), &s2), &[FrLocal::from(2usize), FrLocal::from(34usize)], proof_ref(&proof)));
    }
    fn s1_ref(s: &SigmaG1<G1Local>) -> &SigmaG1<G1Local> { s }
    fn proof_ref(p: &Proof<G1Local, G2Local>) -> &Proof<G1Local, G2Local> { p }
}
