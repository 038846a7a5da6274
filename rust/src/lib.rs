//! ark-pcd-hip: `HipGroth16<E>` -- a `SNARK` whose key / proof types are those of `ark_groth16::Groth16<E>` and
//! whose `prove` runs upstream constraint synthesis and then ONE call into libpcdhip.so (include/pcdhip.h).
//! Drop it into `ECCyclePCDConfig::{MainSNARK, HelpSNARK}` (reference: src/ec_cycle_pcd/mod.rs:24-33).
//! Source only (no Rust toolchain in the build container); see INTEGRATION.md.
use ark_ec::{AffineCurve, PairingEngine};
use ark_ff::{PrimeField, UniformRand};
use ark_groth16::{Groth16, Proof, ProvingKey};
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystem, OptimizationGoal, SynthesisError};
use ark_snark::{CircuitSpecificSetupSNARK, SNARK};
use ark_std::marker::PhantomData;
use ark_std::rand::{CryptoRng, RngCore};

pub mod ffi;

/// Curves the library supports; `CURVE_ID` is `PCDHIP_MNT4_298` ... (include/pcdhip.h).
pub trait HipCurve: PairingEngine {
    const CURVE_ID: u32;
}
impl HipCurve for ark_mnt4_298::MNT4_298 { const CURVE_ID: u32 = 0; }
impl HipCurve for ark_mnt6_298::MNT6_298 { const CURVE_ID: u32 = 1; }

pub struct HipGroth16<E: HipCurve>(PhantomData<E>);

impl<E: HipCurve> SNARK<E::Fr> for HipGroth16<E> {
    type ProvingKey = ProvingKey<E>;
    type VerifyingKey = <Groth16<E> as SNARK<E::Fr>>::VerifyingKey;
    type Proof = Proof<E>;
    type ProcessedVerifyingKey = <Groth16<E> as SNARK<E::Fr>>::ProcessedVerifyingKey;
    type Error = SynthesisError;

    fn circuit_specific_setup<C: ConstraintSynthesizer<E::Fr>, R: RngCore + CryptoRng>(
        circuit: C, rng: &mut R,
    ) -> Result<(Self::ProvingKey, Self::VerifyingKey), Self::Error> {
        <Groth16<E> as CircuitSpecificSetupSNARK<E::Fr>>::setup(circuit, rng)
    }

    fn prove<C: ConstraintSynthesizer<E::Fr>, R: RngCore>(
        pk: &Self::ProvingKey, circuit: C, rng: &mut R,
    ) -> Result<Self::Proof, Self::Error> {
        // same random draws, in the same order, as ark-groth16 `create_random_proof`
        let r = E::Fr::rand(rng);
        let s = E::Fr::rand(rng);
        let cs = ConstraintSystem::new_ref();
        cs.set_optimization_goal(OptimizationGoal::Constraints);
        circuit.generate_constraints(cs.clone())?;
        debug_assert!(cs.is_satisfied().unwrap());
        cs.finalize();
        let matrices = cs.to_matrices().ok_or(SynthesisError::AssignmentMissing)?;
        let prover = cs.borrow().unwrap();
        let mut z: Vec<E::Fr> = prover.instance_assignment.clone();
        z.extend_from_slice(&prover.witness_assignment);
        match ffi::groth16_prove::<E>(pk, &matrices, &z, r, s) {
            Ok(proof) => Ok(proof),
            // domain needs mixed radix (help field above its 2-adicity): run the upstream CPU prover
            Err(ffi::Error::SizeUnsupported) => ffi::cpu_prove_with_rs::<E>(pk, &matrices, &z, r, s),
            Err(_) => Err(SynthesisError::UnexpectedIdentity),
        }
    }

    fn process_vk(vk: &Self::VerifyingKey) -> Result<Self::ProcessedVerifyingKey, Self::Error> {
        Groth16::<E>::process_vk(vk)
    }
    fn verify_with_processed_vk(
        pvk: &Self::ProcessedVerifyingKey, x: &[E::Fr], proof: &Self::Proof,
    ) -> Result<bool, Self::Error> {
        Groth16::<E>::verify_with_processed_vk(pvk, x, proof)
    }
}

/// In-memory image of a field element: `BigInteger` limbs of the Montgomery representation.
pub(crate) fn limbs_of<F: PrimeField>(x: &F) -> &[u64] {
    // ark-ff 0.2/0.3: `Fp320(pub BigInteger320, PhantomData)`; the first field is the Montgomery residue
    unsafe { core::slice::from_raw_parts(x as *const F as *const u64, (F::size_in_bits() + 63) / 64) }
}

pub(crate) fn pack_affine<G: AffineCurve>(pts: &[G], words_per_point: usize) -> (Vec<u64>, Vec<u8>) {
    let mut xy = Vec::with_capacity(pts.len() * words_per_point);
    let mut inf = Vec::with_capacity(pts.len());
    for p in pts {
        inf.push(p.is_zero() as u8);
        ffi::push_point_limbs(p, &mut xy, words_per_point);
    }
    (xy, inf)
}
