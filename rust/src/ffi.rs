//! `extern "C"` binding of include/pcdhip.h (one declaration per entry point the shim uses; the header cites, for each, the
//! upstream function it replaces).  Source only; see INTEGRATION.md.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct pcdhip_ctx { _p: [u8; 0] }
#[repr(C)] pub struct pcdhip_bases { _p: [u8; 0] }
#[repr(C)] pub struct pcdhip_buf { _p: [u8; 0] }
#[repr(C)] pub struct pcdhip_g16_pk { _p: [u8; 0] }
#[repr(C)] pub struct pcdhip_pvk { _p: [u8; 0] }
#[repr(C)] pub struct pcdhip_csr { pub num_rows: u64, pub row_ptr: *const u64, pub col: *const u32, pub coeff: *const u64 }
#[repr(C)]
pub struct pcdhip_g16_pk_host {
    pub curve_id: u32, pub _pad: u32, pub num_vars: u64, pub num_inputs: u64, pub domain_size: u64,
    pub alpha_g1: *const u64, pub beta_g1: *const u64, pub delta_g1: *const u64, pub beta_g2: *const u64, pub delta_g2: *const u64,
    pub a_query: *const u64, pub a_inf: *const u8, pub b_g1_query: *const u64, pub b_g1_inf: *const u8,
    pub b_g2_query: *const u64, pub b_g2_inf: *const u8, pub h_query: *const u64, pub h_inf: *const u8, pub h_len: u64,
    pub l_query: *const u64, pub l_inf: *const u8, pub l_len: u64,
}
#[repr(C)]
pub struct pcdhip_g16_setup_out {
    pub alpha_g1: *mut u64, pub beta_g1: *mut u64, pub delta_g1: *mut u64, pub beta_g2: *mut u64, pub gamma_g2: *mut u64, pub delta_g2: *mut u64,
    pub a_query: *mut u64, pub a_inf: *mut u8, pub b_g1_query: *mut u64, pub b_g1_inf: *mut u8, pub b_g2_query: *mut u64, pub b_g2_inf: *mut u8,
    pub h_query: *mut u64, pub h_inf: *mut u8, pub l_query: *mut u64, pub l_inf: *mut u8, pub gamma_abc_g1: *mut u64, pub gamma_abc_inf: *mut u8,
    pub domain_size: u64,
}

extern "C" {
    pub fn pcdhip_strerror(code: c_int) -> *const c_char;
    pub fn pcdhip_device_count() -> c_int;
    pub fn pcdhip_init(device_id: c_int, out: *mut *mut pcdhip_ctx) -> c_int;
    pub fn pcdhip_init_devices(device_ids: *const c_int, n_dev: c_int, out: *mut *mut pcdhip_ctx) -> c_int;
    pub fn pcdhip_destroy(ctx: *mut pcdhip_ctx);
    pub fn pcdhip_domain_size(field_id: c_int, min_size: usize) -> usize;
    pub fn pcdhip_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn pcdhip_host_free(p: *mut c_void);
    // K3 / K4 / K7: VariableBaseMSM::multi_scalar_mul over resident bases
    pub fn pcdhip_bases_upload(ctx: *mut pcdhip_ctx, curve: c_int, group: c_int, xy: *const u64, inf: *const u8, n: usize, out: *mut *mut pcdhip_bases) -> c_int;
    pub fn pcdhip_bases_free(ctx: *mut pcdhip_ctx, b: *mut pcdhip_bases);
    pub fn pcdhip_set_precompute_budget(ctx: *mut pcdhip_ctx, bytes: usize) -> c_int;
    pub fn pcdhip_get_precompute_budget(ctx: *mut pcdhip_ctx, bytes: *mut usize) -> c_int;
    pub fn pcdhip_msm(ctx: *mut pcdhip_ctx, bases: *const pcdhip_bases, offset: usize, scalars: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn pcdhip_to_affine(ctx: *mut pcdhip_ctx, curve: c_int, group: c_int, xyz: *const u64, n: usize, out_xy: *mut u64, out_inf: *mut u8) -> c_int;
    // K2: Radix2EvaluationDomain / GeneralEvaluationDomain transforms
    pub fn pcdhip_fft(ctx: *mut pcdhip_ctx, field: c_int, data: *mut u64, log_n: u32, inverse: c_int, coset: c_int) -> c_int;
    pub fn pcdhip_fft_general(ctx: *mut pcdhip_ctx, field: c_int, data: *mut u64, n: usize, inverse: c_int, coset: c_int) -> c_int;
    pub fn pcdhip_fft_seq(ctx: *mut pcdhip_ctx, field: c_int, data: *mut u64, n: usize, ops: *const c_int, n_ops: c_int) -> c_int;
    // K1 + K3 + K4 + K5: create_proof after synthesis
    pub fn pcdhip_g16_pk_upload(ctx: *mut pcdhip_ctx, host: *const pcdhip_g16_pk_host, out: *mut *mut pcdhip_g16_pk) -> c_int;
    pub fn pcdhip_g16_pk_set_r1cs(ctx: *mut pcdhip_ctx, pk: *mut pcdhip_g16_pk, a: *const pcdhip_csr, b: *const pcdhip_csr, c: *const pcdhip_csr) -> c_int;
    pub fn pcdhip_g16_pk_free(ctx: *mut pcdhip_ctx, pk: *mut pcdhip_g16_pk);
    /// device bytes a resident key holds: out[0] ordinary copies, out[1] the second layout, out[2] / out[3] copies per point (what a host
    /// that caches several keys budgets with: `prover::MAX_CACHED_KEYS`)
    pub fn pcdhip_g16_pk_memory(pk: *const pcdhip_g16_pk, out: *mut u64) -> c_int;
    pub fn pcdhip_groth16_prove(ctx: *mut pcdhip_ctx, pk: *const pcdhip_g16_pk, a: *const pcdhip_csr, b: *const pcdhip_csr, c: *const pcdhip_csr,
                                z: *const u64, r: *const u64, s: *const u64, proof: *mut u64, inf: *mut u8) -> c_int;
    // key memory: the second layout of a key's assignment queries (a window per proof; INTEGRATION.md "Key memory"): -1 automatic, 0 never, 6..22 bits
    pub fn pcdhip_groth16_set_sparse_window(ctx: *mut pcdhip_ctx, bits: c_int) -> c_int;
    // 8f rank 2: generate_parameters after synthesis
    pub fn pcdhip_groth16_setup(ctx: *mut pcdhip_ctx, curve: c_int, a: *const pcdhip_csr, b: *const pcdhip_csr, c: *const pcdhip_csr,
                                num_vars: usize, num_inputs: usize, g1_xy: *const u64, g2_xy: *const u64, toxic: *const u64,
                                out: *mut pcdhip_g16_setup_out) -> c_int;
    // K6 + 8f rank 3: process_vk, verify_with_processed_vk (n proofs), random-linear-combination batch
    pub fn pcdhip_process_vk(ctx: *mut pcdhip_ctx, curve: c_int, alpha_g1: *const u64, beta_g2: *const u64, gamma_g2: *const u64, delta_g2: *const u64,
                             gamma_abc_g1: *const u64, gamma_abc_inf: *const u8, num_inputs: usize, out: *mut *mut pcdhip_pvk) -> c_int;
    pub fn pcdhip_pvk_free(ctx: *mut pcdhip_ctx, pvk: *mut pcdhip_pvk);
    pub fn pcdhip_groth16_verify_prepared(ctx: *mut pcdhip_ctx, pvk: *const pcdhip_pvk, n_proofs: usize, public_inputs: *const u64,
                                          proofs: *const u64, proofs_inf: *const u8, ok: *mut c_int) -> c_int;
    pub fn pcdhip_groth16_verify_batch_rlc(ctx: *mut pcdhip_ctx, pvk: *const pcdhip_pvk, n_proofs: usize, public_inputs: *const u64,
                                           proofs: *const u64, proofs_inf: *const u8, rho: *const u64, all_ok: *mut c_int) -> c_int;
}

/// `PCDHIP_E_*` (include/pcdhip.h)
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Error { Arg, SizeUnsupported, NoDevice, Oom, Hip }
pub fn check(rc: c_int) -> Result<(), Error> {
    match rc { 0 => Ok(()), -1 => Err(Error::Arg), -2 => Err(Error::SizeUnsupported), -3 => Err(Error::NoDevice), -4 => Err(Error::Oom), _ => Err(Error::Hip) }
}
impl core::fmt::Display for Error {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result { write!(f, "pcdhip: {:?}", self) }
}
impl std::error::Error for Error {}
