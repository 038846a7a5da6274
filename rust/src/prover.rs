//! The calls into libpcdhip.so around one proof, and the CPU route for what the library refuses.
use crate::ffi::{self, Error};
use crate::marshal::{self, Csr};
use crate::HipCurve;
use ark_ec::msm::VariableBaseMSM;
use ark_ec::{AffineCurve, ProjectiveCurve};
use ark_ff::{PrimeField, Zero};
use ark_groth16::r1cs_to_qap::{LibsnarkReduction, R1CSToQAP};
use ark_groth16::{Proof, ProvingKey, VerifyingKey};
use ark_poly::GeneralEvaluationDomain;
use ark_relations::r1cs::{ConstraintMatrices, ConstraintSystemRef, SynthesisError};
use ark_serialize::CanonicalSerialize;
use std::os::raw::c_int;
use std::sync::Mutex;

/// One library context per process and device list (PCDHIP_DEVICES="0,1,..": all of them serve every proof, sharded by point
/// range; default "0"); device keys are cached next to it.  A key is identified by its IDENTITY BYTES -- the compressed
/// serialisation of its verifying key plus the lengths of its queries -- never by its address (a dropped key's address can be
/// reused by a different key) and never by a digest alone: the digest (FNV-1a, not collision resistant) only nominates a cache
/// entry, the hit is confirmed by comparing the identity bytes in full (a colliding key registered first must not
/// capture later lookups -- the hit decides which key a proof is made or checked under).
pub(crate) struct DeviceKey { digest: [u8; 32], ident: Vec<u8>, curve: u32, handle: *mut ffi::pcdhip_g16_pk, has_r1cs: bool, last_use: u64 }
/// a prepared verifying key on the device (`pcdhip_process_vk`: e(alpha, beta), the negated gamma / delta, the window tables of
/// gamma_abc_g1): a PCD verifies every message under the same key, so it is kept -- up to `MAX_CACHED_KEYS` of them, the least
/// recently used one is released with `pcdhip_pvk_free` when another arrives
pub(crate) struct DeviceVk { digest: [u8; 32], ident: Vec<u8>, curve: u32, handle: *mut ffi::pcdhip_pvk, last_use: u64 }
pub(crate) struct Device {
    pub(crate) ctx: *mut ffi::pcdhip_ctx, keys: Vec<DeviceKey>, vks: Vec<DeviceVk>, pub(crate) clock: u64,
    #[cfg(feature = "s2")] pub(crate) bases: Vec<crate::s2::Resident>,
    /// keys of the resident vectors' running digests (rust/src/s2.rs `running_digest`): drawn once per process
    #[cfg(feature = "s2")] pub(crate) digest_keys: [std::collections::hash_map::RandomState; 2],
}
unsafe impl Send for Device {}
static DEVICE: Mutex<Option<Device>> = Mutex::new(None);
/// proving keys / prepared verifying keys kept resident at most (a PCD has two of each; a proving key with its window-shifted
/// copies is gigabytes of HBM)
pub const MAX_CACHED_KEYS: usize = 8;

pub(crate) fn with_device<T>(f: impl FnOnce(&mut Device) -> Result<T, Error>) -> Result<T, Error> {
    let mut guard = DEVICE.lock().unwrap();
    if guard.is_none() {
        let ids: Vec<c_int> = std::env::var("PCDHIP_DEVICES").unwrap_or_else(|_| "0".into())
            .split(',').filter_map(|t| t.trim().parse().ok()).collect();
        let mut ctx = core::ptr::null_mut();
        ffi::check(unsafe { ffi::pcdhip_init_devices(ids.as_ptr(), ids.len() as c_int, &mut ctx) })?;
        // the second layout of a key's assignment queries (a window per proof: the assignment of the reference's circuits is bit decompositions,
        // data_structures.rs:269-304): opt-in (PCDHIP_SPARSE_WINDOW=-1 for the automatic rule, 6..22 for a window outright) -- this host caches up to
        // MAX_CACHED_KEYS keys on one device and the second layout more than doubles a key's memory, so the library leaves it off by default
        if let Some(bits) = std::env::var("PCDHIP_SPARSE_WINDOW").ok().and_then(|v| v.trim().parse::<c_int>().ok()) {
            ffi::check(unsafe { ffi::pcdhip_groth16_set_sparse_window(ctx, bits) })?;
        }
        *guard = Some(Device { ctx, keys: Vec::new(), vks: Vec::new(), clock: 0, #[cfg(feature = "s2")] bases: Vec::new(),
                               #[cfg(feature = "s2")] digest_keys: [std::collections::hash_map::RandomState::new(), std::collections::hash_map::RandomState::new()] });
    }
    f(guard.as_mut().unwrap())
}

/// FNV-1a, four lanes: a cache NOMINATOR, not an identity (see `DeviceKey`)
fn digest_of(bytes: &[u8]) -> [u8; 32] {
    let mut out = [0u8; 32];
    for lane in 0..4u64 {
        let mut h: u64 = 0xcbf29ce484222325 ^ lane.wrapping_mul(0x9e3779b97f4a7c15);
        for b in bytes { h ^= *b as u64; h = h.wrapping_mul(0x100000001b3); }
        out[lane as usize * 8..][..8].copy_from_slice(&h.to_le_bytes());
    }
    out
}

/// What identifies a proving key: its verifying key (compressed serialisation) and the lengths of its queries.
fn key_ident<E: HipCurve>(pk: &ProvingKey<E>) -> Vec<u8> {
    let mut bytes = Vec::new();
    pk.vk.serialize(&mut bytes).expect("serialising into a Vec cannot fail");
    for n in &[pk.a_query.len(), pk.b_g1_query.len(), pk.b_g2_query.len(), pk.h_query.len(), pk.l_query.len()] {
        bytes.extend_from_slice(&(*n as u64).to_le_bytes());
    }
    bytes
}

fn pack_g1<E: HipCurve>(pts: &[E::G1Affine]) -> (Vec<u64>, Vec<u8>) {
    let mut xy = Vec::with_capacity(pts.len() * 2 * E::FQ_LIMBS);
    let mut inf = Vec::with_capacity(pts.len());
    for p in pts { E::push_g1(p, &mut xy, &mut inf); }
    (xy, inf)
}
fn pack_g2<E: HipCurve>(pts: &[E::G2Affine]) -> (Vec<u64>, Vec<u8>) {
    let mut xy = Vec::with_capacity(pts.len() * 2 * E::G2_DEG * E::FQ_LIMBS);
    let mut inf = Vec::with_capacity(pts.len());
    for p in pts { E::push_g2(p, &mut xy, &mut inf); }
    (xy, inf)
}

/// Upload `pk` (once per key: the library also builds its window-shifted copies) and return the cached handle.
fn device_key<E: HipCurve>(dev: &mut Device, pk: &ProvingKey<E>, num_vars: usize, num_inputs: usize, domain: usize) -> Result<usize, Error> {
    let ident = key_ident::<E>(pk);
    let digest = digest_of(&ident);
    dev.clock += 1;
    let now = dev.clock;
    if let Some(i) = dev.keys.iter().position(|k| k.digest == digest && k.curve == E::CURVE_ID && k.ident == ident) {
        dev.keys[i].last_use = now;
        return Ok(i);
    }
    if dev.keys.len() >= MAX_CACHED_KEYS {
        let victim = dev.keys.iter().enumerate().min_by_key(|(_, k)| k.last_use).map(|(i, _)| i).unwrap();
        let old = dev.keys.swap_remove(victim);
        unsafe { ffi::pcdhip_g16_pk_free(dev.ctx, old.handle) };
    }
    let one = |p: &E::G1Affine| pack_g1::<E>(core::slice::from_ref(p)).0;
    let one2 = |p: &E::G2Affine| pack_g2::<E>(core::slice::from_ref(p)).0;
    let (alpha, beta1, delta1) = (one(&pk.vk.alpha_g1), one(&pk.beta_g1), one(&pk.delta_g1));
    let (beta2, delta2) = (one2(&pk.vk.beta_g2), one2(&pk.vk.delta_g2));
    let (a, a_inf) = pack_g1::<E>(&pk.a_query);
    let (b1, b1_inf) = pack_g1::<E>(&pk.b_g1_query);
    let (b2, b2_inf) = pack_g2::<E>(&pk.b_g2_query);
    let (h, h_inf) = pack_g1::<E>(&pk.h_query);
    let (l, l_inf) = pack_g1::<E>(&pk.l_query);
    if pk.a_query.len() != num_vars || pk.l_query.len() != num_vars - num_inputs { return Err(Error::Arg); }
    let host = ffi::pcdhip_g16_pk_host {
        curve_id: E::CURVE_ID, _pad: 0, num_vars: num_vars as u64, num_inputs: num_inputs as u64, domain_size: domain as u64,
        alpha_g1: alpha.as_ptr(), beta_g1: beta1.as_ptr(), delta_g1: delta1.as_ptr(), beta_g2: beta2.as_ptr(), delta_g2: delta2.as_ptr(),
        a_query: a.as_ptr(), a_inf: a_inf.as_ptr(), b_g1_query: b1.as_ptr(), b_g1_inf: b1_inf.as_ptr(),
        b_g2_query: b2.as_ptr(), b_g2_inf: b2_inf.as_ptr(), h_query: h.as_ptr(), h_inf: h_inf.as_ptr(), h_len: pk.h_query.len() as u64,
        l_query: l.as_ptr(), l_inf: l_inf.as_ptr(), l_len: pk.l_query.len() as u64,
    };
    let mut handle = core::ptr::null_mut();
    ffi::check(unsafe { ffi::pcdhip_g16_pk_upload(dev.ctx, &host, &mut handle) })?;
    dev.keys.push(DeviceKey { digest, ident, curve: E::CURVE_ID, handle, has_r1cs: false, last_use: now });
    Ok(dev.keys.len() - 1)
}

/// `create_proof` after synthesis, on the device: witness map + five MSMs + assembly with the caller's `r`, `s`.
pub fn groth16_prove<E: HipCurve>(
    pk: &ProvingKey<E>, m: &ConstraintMatrices<E::Fr>, z: &[E::Fr], r: E::Fr, s: E::Fr,
) -> Result<Proof<E>, Error> {
    let num_inputs = m.num_instance_variables;
    let num_vars = m.num_instance_variables + m.num_witness_variables;
    if z.len() != num_vars { return Err(Error::Arg); }
    let field_id = if E::CURVE_ID % 2 == 0 { E::CURVE_ID + 1 } else { E::CURVE_ID - 1 };  // Fr of MNT4 = Fq of MNT6 and vice versa
    let domain = unsafe { ffi::pcdhip_domain_size(field_id as c_int, m.num_constraints + num_inputs) };
    if domain == 0 { return Err(Error::SizeUnsupported); }
    with_device(|dev| {
        let i = device_key::<E>(dev, pk, num_vars, num_inputs, domain)?;
        if !dev.keys[i].has_r1cs {  // the matrices are fixed per circuit, like the key: resident from the first proof on
            let (a, b, c) = (Csr::from_matrix(&m.a), Csr::from_matrix(&m.b), Csr::from_matrix(&m.c));
            ffi::check(unsafe { ffi::pcdhip_g16_pk_set_r1cs(dev.ctx, dev.keys[i].handle, &a.view(), &b.view(), &c.view()) })?;
            dev.keys[i].has_r1cs = true;
        }
        let mut zl = Vec::with_capacity(z.len() * ((E::Fr::size_in_bits() + 63) / 64));
        for v in z { marshal::push_fp(v, &mut zl); }
        let (rl, sl) = (marshal::limbs_of(&r).to_vec(), marshal::limbs_of(&s).to_vec());
        let (w1, w2) = (2 * E::FQ_LIMBS, 2 * E::G2_DEG * E::FQ_LIMBS);
        let mut proof = vec![0u64; 2 * w1 + w2];
        let mut inf = [0u8; 3];
        ffi::check(unsafe {
            ffi::pcdhip_groth16_prove(dev.ctx, dev.keys[i].handle, core::ptr::null(), core::ptr::null(), core::ptr::null(),
                                      zl.as_ptr(), rl.as_ptr(), sl.as_ptr(), proof.as_mut_ptr(), inf.as_mut_ptr())
        })?;
        Ok(Proof {
            a: E::g1_from(&proof[..w1], inf[0] != 0),
            b: E::g2_from(&proof[w1..w1 + w2], inf[1] != 0),
            c: E::g1_from(&proof[w1 + w2..], inf[2] != 0),
        })
    })
}

/// The same arithmetic on the CPU with upstream's own primitives, from the constraint system that was already synthesised (the
/// circuit has been consumed; `SNARK::prove` has no `Clone` bound): the body of ark-groth16 `create_proof` after synthesis.
pub fn cpu_prove_with_rs<E: HipCurve>(
    pk: &ProvingKey<E>, cs: ConstraintSystemRef<E::Fr>, r: E::Fr, s: E::Fr,
) -> Result<Proof<E>, SynthesisError> {
    let h = LibsnarkReduction::witness_map::<E::Fr, GeneralEvaluationDomain<E::Fr>>(cs.clone())?;
    let prover = cs.borrow().ok_or(SynthesisError::MissingCS)?;
    let repr = |v: &[E::Fr]| v.iter().map(|x| x.into_repr()).collect::<Vec<_>>();
    let h_acc = VariableBaseMSM::multi_scalar_mul(&pk.h_query, &repr(&h));
    let aux = repr(&prover.witness_assignment);
    let l_aux_acc = VariableBaseMSM::multi_scalar_mul(&pk.l_query, &aux);
    let mut assignment = repr(&prover.instance_assignment[1..]);
    assignment.extend_from_slice(&aux);
    let r_s_delta_g1 = pk.delta_g1.into_projective().mul(r.into_repr()).mul(s.into_repr());
    let coeff1 = |init: E::G1Projective, q: &[E::G1Affine], vk_param: E::G1Affine| {
        let mut res = init;
        res.add_assign_mixed(&q[0]);
        res += &VariableBaseMSM::multi_scalar_mul(&q[1..], &assignment);
        res.add_assign_mixed(&vk_param);
        res
    };
    let g_a = coeff1(pk.delta_g1.mul(r), &pk.a_query, pk.vk.alpha_g1);
    let g1_b = if r.is_zero() { E::G1Projective::zero() } else { coeff1(pk.delta_g1.mul(s), &pk.b_g1_query, pk.beta_g1) };
    let g2_b = {
        let mut res = pk.vk.delta_g2.mul(s);
        res.add_assign_mixed(&pk.b_g2_query[0]);
        res += &VariableBaseMSM::multi_scalar_mul(&pk.b_g2_query[1..], &assignment);
        res.add_assign_mixed(&pk.vk.beta_g2);
        res
    };
    let mut g_c = g_a.mul(s.into_repr());
    g_c += &g1_b.mul(r.into_repr());
    g_c -= &r_s_delta_g1;
    g_c += &l_aux_acc;
    g_c += &h_acc;
    Ok(Proof { a: g_a.into_affine(), b: g2_b.into_affine(), c: g_c.into_affine() })
}

/// `ECCyclePCD::verify` for every prior message of a merge node in one call (mod.rs:239 once per input): `process_vk` on the
/// device, then either per-proof answers or, with `rho` (one non-zero 128-bit challenge per proof from the caller's RNG), a
/// single product with one shared final exponentiation.
pub fn verify_batch<E: HipCurve>(
    vk: &VerifyingKey<E>, inputs: &[Vec<E::Fr>], proofs: &[Proof<E>], rho: Option<&[[u64; 2]]>,
) -> Result<Vec<bool>, Error> {
    let n = proofs.len();
    if inputs.len() != n || inputs.iter().any(|x| x.len() + 1 != vk.gamma_abc_g1.len()) { return Err(Error::Arg); }
    with_device(|dev| {
        let one = |p: &E::G1Affine| pack_g1::<E>(core::slice::from_ref(p)).0;
        let one2 = |p: &E::G2Affine| pack_g2::<E>(core::slice::from_ref(p)).0;
        // the prepared key is made once per verifying key and kept: `process_vk` costs a pairing and the window tables, a prepared
        // verification 5 ms (MNT4-298) / 24 ms (MNT4-753) for one proof or sixty-four.  Identity = the compressed serialisation
        // of the key, compared IN FULL on a digest hit.
        let mut ident = Vec::new();
        vk.serialize(&mut ident).expect("serialising into a Vec cannot fail");
        let digest = digest_of(&ident);
        dev.clock += 1;
        let now = dev.clock;
        let pvk = match dev.vks.iter_mut().find(|k| k.digest == digest && k.curve == E::CURVE_ID && k.ident == ident) {
            Some(k) => { k.last_use = now; k.handle }
            None => {
                if dev.vks.len() >= MAX_CACHED_KEYS {
                    let victim = dev.vks.iter().enumerate().min_by_key(|(_, k)| k.last_use).map(|(i, _)| i).unwrap();
                    let old = dev.vks.swap_remove(victim);
                    unsafe { ffi::pcdhip_pvk_free(dev.ctx, old.handle) };
                }
                let (abc, abc_inf) = pack_g1::<E>(&vk.gamma_abc_g1);
                let mut pvk = core::ptr::null_mut();
                ffi::check(unsafe {
                    ffi::pcdhip_process_vk(dev.ctx, E::CURVE_ID as c_int, one(&vk.alpha_g1).as_ptr(), one2(&vk.beta_g2).as_ptr(), one2(&vk.gamma_g2).as_ptr(),
                                           one2(&vk.delta_g2).as_ptr(), abc.as_ptr(), abc_inf.as_ptr(), vk.gamma_abc_g1.len(), &mut pvk)
                })?;
                dev.vks.push(DeviceVk { digest, ident, curve: E::CURVE_ID, handle: pvk, last_use: now });
                pvk
            }
        };
        let mut pubs = Vec::new();
        for x in inputs { for v in x { marshal::push_repr(v, &mut pubs); } }
        let (mut pr, mut pr_inf) = (Vec::new(), Vec::new());
        for p in proofs { E::push_g1(&p.a, &mut pr, &mut pr_inf); E::push_g2(&p.b, &mut pr, &mut pr_inf); E::push_g1(&p.c, &mut pr, &mut pr_inf); }
        let res = match rho {
            Some(rho) if rho.len() == n => {
                let mut all = 0 as c_int;
                let flat: Vec<u64> = rho.iter().flat_map(|r| r.iter().copied()).collect();
                ffi::check(unsafe { ffi::pcdhip_groth16_verify_batch_rlc(dev.ctx, pvk, n, pubs.as_ptr(), pr.as_ptr(), pr_inf.as_ptr(), flat.as_ptr(), &mut all) })
                    .map(|_| vec![all == 1; n])
            }
            Some(_) => Err(Error::Arg),
            None => {
                let mut ok = vec![0 as c_int; n];
                ffi::check(unsafe { ffi::pcdhip_groth16_verify_prepared(dev.ctx, pvk, n, pubs.as_ptr(), pr.as_ptr(), pr_inf.as_ptr(), ok.as_mut_ptr()) })
                    .map(|_| ok.iter().map(|v| *v == 1).collect())
            }
        };
        res
    })
}
