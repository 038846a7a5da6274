//! ark-pcd-hip: `HipGroth16<E>` -- a `SNARK` whose key / proof types are those of `ark_groth16::Groth16<E>` and whose `prove`
//! runs upstream constraint synthesis and then ONE call into libpcdhip.so (include/pcdhip.h) -- plus the matching verifier
//! gadget, so that both can be dropped into `ECCyclePCDConfig::{MainSNARK, HelpSNARK, MainSNARKGadget, HelpSNARKGadget}`
//! (reference: src/ec_cycle_pcd/mod.rs:24-33; usage mirrors tests/mnt4_groth16.rs:22-30 with the type names swapped).
//! Source only (no Rust toolchain in the build container: never compiled); see INTEGRATION.md.
use ark_crypto_primitives::snark::constraints::SNARKGadget;
use ark_ec::PairingEngine;
use ark_ff::{UniformRand, Zero};
use ark_groth16::{constraints::Groth16VerifierGadget, Groth16, Proof, ProvingKey, VerifyingKey};
use ark_r1cs_std::{boolean::Boolean, pairing::PairingVar};
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystem, OptimizationGoal, SynthesisError};
use ark_snark::{CircuitSpecificSetupSNARK, SNARK};
use ark_std::marker::PhantomData;
use ark_std::rand::{CryptoRng, RngCore};

pub mod ffi;
pub mod marshal;
pub mod prover;
#[cfg(feature = "s2")]
pub mod s2;

/// Curves the library supports.  `GroupAffine` exposes its coordinates only on the concrete type, so each curve says how its
/// points turn into C-ABI limbs (x || y, extension coefficients c0, c1 (, c2) in order) and back.
pub trait HipCurve: PairingEngine {
    /// `PCDHIP_MNT4_298` ... (include/pcdhip.h)
    const CURVE_ID: u32;
    /// u64 limbs of one base-field element (5 / 12)
    const FQ_LIMBS: usize;
    /// base-field coefficients per G2 coordinate (2: MNT4 twist over Fq2, 3: MNT6 twist over Fq3)
    const G2_DEG: usize;
    fn push_g1(p: &Self::G1Affine, xy: &mut Vec<u64>, inf: &mut Vec<u8>);
    fn push_g2(p: &Self::G2Affine, xy: &mut Vec<u64>, inf: &mut Vec<u8>);
    fn g1_from(xy: &[u64], inf: bool) -> Self::G1Affine;
    fn g2_from(xy: &[u64], inf: bool) -> Self::G2Affine;
    /// the library's Jacobian X || Y || Z (Z = 0: the identity) as the group's projective type (seam S2: an MSM result)
    fn g1_projective(xyz: &[u64]) -> Self::G1Projective;
    fn g2_projective(xyz: &[u64]) -> Self::G2Projective;
}

macro_rules! impl_hip_curve {
    ($engine:ty, $krate:ident, $id:expr, $limbs:expr, $deg:expr, [$($c:ident),+]) => {
        impl HipCurve for $engine {
            const CURVE_ID: u32 = $id;
            const FQ_LIMBS: usize = $limbs;
            const G2_DEG: usize = $deg;
            fn push_g1(p: &Self::G1Affine, xy: &mut Vec<u64>, inf: &mut Vec<u8>) {
                inf.push(p.infinity as u8);
                // (a flagged point's coordinates are ignored by the library: it rewrites them to its own encoding of infinity)
                marshal::push_fp(&p.x, xy);
                marshal::push_fp(&p.y, xy);
            }
            fn push_g2(p: &Self::G2Affine, xy: &mut Vec<u64>, inf: &mut Vec<u8>) {
                inf.push(p.infinity as u8);
                $( marshal::push_fp(&p.x.$c, xy); )+
                $( marshal::push_fp(&p.y.$c, xy); )+
            }
            fn g1_from(xy: &[u64], inf: bool) -> Self::G1Affine {
                if inf { return <Self::G1Affine as Zero>::zero(); }
                let l = $limbs;
                $krate::G1Affine::new(marshal::fp_from_limbs(&xy[..l]), marshal::fp_from_limbs(&xy[l..2 * l]), false)
            }
            fn g2_from(xy: &[u64], inf: bool) -> Self::G2Affine {
                if inf { return <Self::G2Affine as Zero>::zero(); }
                let l = $limbs;
                let mut k = 0usize;
                let mut next = || { let v = marshal::fp_from_limbs(&xy[k * l..(k + 1) * l]); k += 1; v };
                let mut x = <$krate::G2Affine as ark_ec::AffineCurve>::BaseField::zero();
                let mut y = x;
                $( x.$c = next(); )+
                $( y.$c = next(); )+
                $krate::G2Affine::new(x, y, false)
            }
            fn g1_projective(w: &[u64]) -> Self::G1Projective {
                let l = $limbs;
                $krate::G1Projective::new(marshal::fp_from_limbs(&w[..l]), marshal::fp_from_limbs(&w[l..2 * l]), marshal::fp_from_limbs(&w[2 * l..3 * l]))
            }
            fn g2_projective(w: &[u64]) -> Self::G2Projective {
                let l = $limbs;
                let mut k = 0usize;
                let mut next = || { let v = marshal::fp_from_limbs(&w[k * l..(k + 1) * l]); k += 1; v };
                let mut x = <$krate::G2Affine as ark_ec::AffineCurve>::BaseField::zero();
                let (mut y, mut z) = (x, x);
                $( x.$c = next(); )+
                $( y.$c = next(); )+
                $( z.$c = next(); )+
                $krate::G2Projective::new(x, y, z)
            }
        }
    };
}
impl_hip_curve!(ark_mnt4_298::MNT4_298, ark_mnt4_298, 0, 5, 2, [c0, c1]);
impl_hip_curve!(ark_mnt6_298::MNT6_298, ark_mnt6_298, 1, 5, 3, [c0, c1, c2]);
impl_hip_curve!(ark_mnt4_753::MNT4_753, ark_mnt4_753, 2, 12, 2, [c0, c1]);
impl_hip_curve!(ark_mnt6_753::MNT6_753, ark_mnt6_753, 3, 12, 3, [c0, c1, c2]);

pub struct HipGroth16<E: HipCurve>(PhantomData<E>);

/// Proofs over fewer constraints than this stay on the CPU (upstream arithmetic on the same synthesis).  Every synthesis of the
/// reference's Main / Help circuit makes a `circuit_specific_setup` + `prove` of a `DefaultCircuit` with `HelpSNARK`
/// (/root/reference src/ec_cycle_pcd/data_structures.rs:134-143, :339-350): a few hundred rows, for which a key digest, an
/// upload and the window-shifted copies of five queries cost far more than the proof.  `PCDHIP_MIN_CONSTRAINTS` overrides.
pub const DEFAULT_MIN_CONSTRAINTS: usize = 1 << 12;
fn min_constraints() -> usize {
    std::env::var("PCDHIP_MIN_CONSTRAINTS").ok().and_then(|v| v.parse().ok()).unwrap_or(DEFAULT_MIN_CONSTRAINTS)
}

impl<E: HipCurve> SNARK<E::Fr> for HipGroth16<E> {
    // identical key / proof types: `ECCyclePCDPK` / `ECCyclePCDVK` (data_structures.rs:14-24, :40-47) are unchanged and keys
    // made by either SNARK work with the other
    type ProvingKey = ProvingKey<E>;
    type VerifyingKey = VerifyingKey<E>;
    type Proof = Proof<E>;
    type ProcessedVerifyingKey = <Groth16<E> as SNARK<E::Fr>>::ProcessedVerifyingKey;
    type Error = SynthesisError;

    fn circuit_specific_setup<C: ConstraintSynthesizer<E::Fr>, R: RngCore + CryptoRng>(
        circuit: C, rng: &mut R,
    ) -> Result<(Self::ProvingKey, Self::VerifyingKey), Self::Error> {
        // (key generation stays upstream here; `pcdhip_groth16_setup` is the device path for it -- INTEGRATION.md -- and keys
        // are the same type either way)
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
        if cs.num_constraints() < min_constraints() {
            // the tiny inner proofs of a synthesis (and anything else below the threshold): no device round trip
            return prover::cpu_prove_with_rs::<E>(pk, cs, r, s);
        }
        let matrices = cs.to_matrices().ok_or(SynthesisError::AssignmentMissing)?;
        let z: Vec<E::Fr> = {
            let prover = cs.borrow().ok_or(SynthesisError::MissingCS)?;
            let mut z = prover.instance_assignment.clone();
            z.extend_from_slice(&prover.witness_assignment);
            z
        };
        match prover::groth16_prove::<E>(pk, &matrices, &z, r, s) {
            Ok(proof) => Ok(proof),
            // a domain the library does not build, or no usable device: the upstream CPU arithmetic on the SAME synthesis
            // (that decision lives here, in the Rust host; the library itself has no CPU path)
            Err(ffi::Error::SizeUnsupported) | Err(ffi::Error::NoDevice) => prover::cpu_prove_with_rs::<E>(pk, cs, r, s),
            Err(_) => Err(SynthesisError::UnexpectedIdentity),
        }
    }

    fn process_vk(vk: &Self::VerifyingKey) -> Result<Self::ProcessedVerifyingKey, Self::Error> { Groth16::<E>::process_vk(vk) }
    fn verify_with_processed_vk(
        pvk: &Self::ProcessedVerifyingKey, x: &[E::Fr], proof: &Self::Proof,
    ) -> Result<bool, Self::Error> {
        // 753-bit curves: the device's prepared verification (31 ms, key prepared once and cached) beats a host core (45 ms); 298-bit:
        // a single verification is as fast on a core (4 ms) -- a merge node's BATCH goes through prover::verify_batch either way
        if E::FQ_LIMBS > 5 {
            if let Ok(ok) = prover::verify_batch::<E>(&pvk.vk, &[x.to_vec()], core::slice::from_ref(proof), None) { return Ok(ok[0]); }
        }
        Groth16::<E>::verify_with_processed_vk(pvk, x, proof)
    }
}

/// `ECCyclePCD<.., IC>: CircuitSpecificSetupPCD` is bounded on `IC::MainSNARK: CircuitSpecificSetupSNARK<MainField>` and
/// `IC::HelpSNARK: CircuitSpecificSetupSNARK<HelpField>` (/root/reference src/ec_cycle_pcd/mod.rs:248-254).  The trait's one
/// method, `setup`, has a default body that calls `SNARK::circuit_specific_setup` -- the marker impl is all it takes.
impl<E: HipCurve> CircuitSpecificSetupSNARK<E::Fr> for HipGroth16<E> {}

/// `ECCyclePCDConfig::{MainSNARKGadget, HelpSNARKGadget}` must implement `SNARKGadget<F, ConstraintF, ThatSNARK>`
/// (mod.rs:31-32) and upstream's gadget is implemented for `Groth16<E>` specifically; this newtype implements it for
/// `HipGroth16<E>` by delegation (the key / proof types, hence all the `Var` types, are the same).
pub struct HipGroth16VerifierGadget<E: HipCurve, P: PairingVar<E, E::Fq>>(PhantomData<(E, P)>);

type Inner<E, P> = Groth16VerifierGadget<E, P>;
type InnerS<E> = Groth16<E>;

impl<E: HipCurve, P: PairingVar<E, E::Fq>> SNARKGadget<E::Fr, E::Fq, HipGroth16<E>> for HipGroth16VerifierGadget<E, P> {
    type ProcessedVerifyingKeyVar = <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::ProcessedVerifyingKeyVar;
    type VerifyingKeyVar = <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::VerifyingKeyVar;
    type InputVar = <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::InputVar;
    type ProofVar = <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::ProofVar;
    type VerifierSize = <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::VerifierSize;

    fn verifier_size(circuit_vk: &VerifyingKey<E>) -> Self::VerifierSize {
        <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::verifier_size(circuit_vk)
    }
    fn verify_with_processed_vk(
        circuit_pvk: &Self::ProcessedVerifyingKeyVar, x: &Self::InputVar, proof: &Self::ProofVar,
    ) -> Result<Boolean<E::Fq>, SynthesisError> {
        <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::verify_with_processed_vk(circuit_pvk, x, proof)
    }
    fn verify(
        circuit_vk: &Self::VerifyingKeyVar, x: &Self::InputVar, proof: &Self::ProofVar,
    ) -> Result<Boolean<E::Fq>, SynthesisError> {
        <Inner<E, P> as SNARKGadget<E::Fr, E::Fq, InnerS<E>>>::verify(circuit_vk, x, proof)
    }
}
