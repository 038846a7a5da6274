//! Known-answer test against the REAL arkworks: computes, with upstream ark-ec / ark-poly / ark-groth16 / ark-serialize, the values
//! this repository's oracle, golden vectors and HIP kernels are pinned to, for the golden INPUTS, and writes them as text.
//!
//!     python tools/kat_export.py                                   # tests/golden/*.npz -> rust/tests/kat_inputs.txt
//!     (cd rust && cargo test --release --test kat -- --nocapture)  # this file -> rust/tests/kat_outputs.txt
//!     python tools/check_kat.py                                    # kat_outputs.txt == tests/golden/*.npz ?
//!
//! Covered: variable-base MSM on the eight groups (`VariableBaseMSM::multi_scalar_mul`), radix-2 transforms on the four scalar
//! fields (`Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}`), the reduced pairing on the four curves, the witness map
//! and the Groth16 proof for given (r, s) on MNT4-298 / MNT6-298 (`create_proof_with_reduction` over a circuit that replays the
//! golden constraint system), and the `CanonicalSerialize` bytes of points.  Never compiled in the build container (no toolchain).
//!
//! Round 4 widens it to what the product only RECALLS of upstream (tools/kat_extra.py says why each matters):
//! `MixedRadixEvaluationDomain` (all four transforms on 7 * 2^3, 49 * 2^2, 5 * 2^3, 25 * 2^2), the witness map over
//! `GeneralEvaluationDomain` past the field's 2-adicity (which domain upstream PICKS, and h on it), `FixedBaseMSM` + batch
//! normalisation on the eight groups, `generate_parameters` with the golden toxic waste (tau injected through the RNG the function
//! samples it from), and the field / group constants themselves.
//!
//! Round 5: every OUTPUT array of tests/golden/*.npz now has a line here (the field operations of fields.npz, the all-ones MSMs, the
//! witness map's h, the serialised proof and verifying key were missing) -- `EMITS` below is the manifest tests/test_kat_bridge.py checks
//! the golden files against, and `kat()` asserts that what was written matches it -- plus two cases for what round 4's infinity skip
//! relies on: `generate_parameters` over a system with variables absent from A / B (WHICH query entries come out as the identity) and
//! `VariableBaseMSM` over bases that are the identity.
use ark_ec::msm::{FixedBaseMSM, VariableBaseMSM};
use ark_ec::{AffineCurve, PairingEngine, ProjectiveCurve};
use ark_ff::{BigInteger, FftField, FftParameters, Field, FpParameters, PrimeField, Zero};
use ark_groth16::{create_proof_with_reduction, generate_parameters, r1cs_to_qap::{LibsnarkReduction, R1CSToQAP}, Proof, ProvingKey, VerifyingKey};
use ark_pcd_hip::{marshal, HipCurve};
use ark_poly::{EvaluationDomain, GeneralEvaluationDomain, MixedRadixEvaluationDomain, Radix2EvaluationDomain};
use ark_relations::lc;
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystem, ConstraintSystemRef, LinearCombination, OptimizationGoal, SynthesisError, Variable};
use ark_serialize::CanonicalSerialize;
use std::collections::HashMap;
use std::fmt::Write as _;

type Arrays = HashMap<String, (Vec<usize>, Vec<u64>)>;

/// Every family of lines this test writes (digits of a name replaced by `{}`): the manifest tests/test_kat_bridge.py compares the OUTPUT
/// arrays of tests/golden/*.npz and tools/kat_extra.expected() with -- a golden array without a line here fails the CPU suite.
const EMITS: &[&str] = &[
    "fields.f{}_add", "fields.f{}_sub", "fields.f{}_mul", "fields.f{}_inv_b", "fields.f{}_a_canonical",
    "msm.c{}_g{}_result_xy", "msm.c{}_g{}_result_inf", "msm.c{}_g{}_ones_xy", "msm.c{}_g{}_ones_inf",
    "fft.f{}_n{}_i{}c{}", "pairing.c{}_gt", "groth16.c{}_h", "groth16.c{}_proof",
    "wire.c{}_g{}_ser{}", "wire.c{}_proof_ser{}", "wire.c{}_vk_ser{}",
    "groth16.c{}_alpha_g{}", "groth16.c{}_beta_g{}", "groth16.c{}_delta_g{}", "groth16.c{}_gamma_g{}",
    "groth16.c{}_a_query", "groth16.c{}_a_query_inf", "groth16.c{}_b_g{}_query", "groth16.c{}_b_g{}_query_inf", "groth16.c{}_h_query",
    "groth16.c{}_h_query_inf", "groth16.c{}_l_query", "groth16.c{}_l_query_inf", "groth16.c{}_gamma_abc_g{}", "groth16.c{}_gamma_abc_g{}_inf",
    "x_mixed.f{}_n{}_i{}c{}", "x_wm.domain_size", "x_wm.h", "x_fixed.c{}_g{}_out_xy", "x_fixed.c{}_g{}_out_inf",
    "x_consts.f{}_generator", "x_consts.f{}_two_adic_root", "x_consts.f{}_two_adicity", "x_consts.f{}_small_subgroup_base",
    "x_consts.f{}_small_subgroup_base_adicity", "x_consts.c{}_g{}_generator",
    "x_setup_inf.a_query", "x_setup_inf.a_query_inf", "x_setup_inf.b_g{}_query", "x_setup_inf.b_g{}_query_inf",
    "x_msm_inf.c{}_g{}_result_xy", "x_msm_inf.c{}_g{}_result_inf",
];
fn family(name: &str) -> String {   // digits after the file prefix -> "{}"
    let dot = name.find('.').map(|i| i + 1).unwrap_or(0);
    let mut o = String::from(&name[..dot]);
    let mut in_digits = false;
    for ch in name[dot..].chars() {
        if ch.is_ascii_digit() { if !in_digits { o.push_str("{}"); in_digits = true; } } else { o.push(ch); in_digits = false; }
    }
    o
}

/// fields.npz: the field operations themselves on the golden operand pairs (Montgomery limbs in and out; `a_canonical` = `into_repr`)
fn fields<F: PrimeField>(a: &Arrays, fid: usize, out: &mut String) {
    let x: Vec<F> = fr_vec(&a[&format!("fields.f{}_a", fid)]);
    let y: Vec<F> = fr_vec(&a[&format!("fields.f{}_b", fid)]);
    let shape = a[&format!("fields.f{}_a", fid)].0.clone();
    let zip = |f: &dyn Fn(&F, &F) -> F| -> Vec<F> { x.iter().zip(y.iter()).map(|(p, q)| f(p, q)).collect() };
    emit(out, &format!("fields.f{}_add", fid), "uint64", &shape, &fr_limbs(&zip(&|p, q| *p + *q)));
    emit(out, &format!("fields.f{}_sub", fid), "uint64", &shape, &fr_limbs(&zip(&|p, q| *p - *q)));
    emit(out, &format!("fields.f{}_mul", fid), "uint64", &shape, &fr_limbs(&zip(&|p, q| *p * *q)));
    emit(out, &format!("fields.f{}_inv_b", fid), "uint64", &shape, &fr_limbs(&zip(&|_, q| q.inverse().unwrap_or_else(F::zero))));
    let canon: Vec<u64> = x.iter().flat_map(|p| p.into_repr().as_ref().to_vec()).collect();
    emit(out, &format!("fields.f{}_a_canonical", fid), "uint64", &shape, &canon);
}

fn msm<E: HipCurve>(a: &Arrays, out: &mut String) {
    let c = E::CURVE_ID;
    let scal = |name: &str| -> Vec<<E::Fr as PrimeField>::BigInt> {
        a[name].1.chunks(a[name].0[1]).map(|l| { let mut b = <E::Fr as PrimeField>::BigInt::default(); b.as_mut().copy_from_slice(l); b }).collect()
    };
    let p1 = g1s::<E>(a, &format!("msm.c{}_g1_bases", c), &format!("msm.c{}_g1_inf", c));
    let r1 = VariableBaseMSM::multi_scalar_mul(&p1, &scal(&format!("msm.c{}_g1_scalars", c))).into_affine();
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    E::push_g1(&r1, &mut xy, &mut inf);
    if r1.is_zero() { xy.iter_mut().for_each(|w| *w = 0); }
    emit(out, &format!("msm.c{}_g1_result_xy", c), "uint64", &[xy.len()], &xy);
    emit(out, &format!("msm.c{}_g1_result_inf", c), "uint8", &[1], &[inf[0] as u64]);
    let one = <E::Fr as PrimeField>::BigInt::from(1u64);   // every scalar equal to one: the branch upstream adds in window 0 only
    let o1 = VariableBaseMSM::multi_scalar_mul(&p1, &vec![one; p1.len()]).into_affine();
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    E::push_g1(&o1, &mut xy, &mut inf);
    if o1.is_zero() { xy.iter_mut().for_each(|w| *w = 0); }
    emit(out, &format!("msm.c{}_g1_ones_xy", c), "uint64", &[xy.len()], &xy);
    emit(out, &format!("msm.c{}_g1_ones_inf", c), "uint8", &[1], &[inf[0] as u64]);
    let p2 = g2s::<E>(a, &format!("msm.c{}_g2_bases", c), &format!("msm.c{}_g2_inf", c));
    let r2 = VariableBaseMSM::multi_scalar_mul(&p2, &scal(&format!("msm.c{}_g2_scalars", c))).into_affine();
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    E::push_g2(&r2, &mut xy, &mut inf);
    if r2.is_zero() { xy.iter_mut().for_each(|w| *w = 0); }
    emit(out, &format!("msm.c{}_g2_result_xy", c), "uint64", &[xy.len()], &xy);
    emit(out, &format!("msm.c{}_g2_result_inf", c), "uint8", &[1], &[inf[0] as u64]);
    let o2 = VariableBaseMSM::multi_scalar_mul(&p2, &vec![one; p2.len()]).into_affine();
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    E::push_g2(&o2, &mut xy, &mut inf);
    if o2.is_zero() { xy.iter_mut().for_each(|w| *w = 0); }
    emit(out, &format!("msm.c{}_g2_ones_xy", c), "uint64", &[xy.len()], &xy);
    emit(out, &format!("msm.c{}_g2_ones_inf", c), "uint8", &[1], &[inf[0] as u64]);
}

/// round 5: `multi_scalar_mul` over bases some of which ARE the identity (`infinity = true`), non-zero scalars on them -- upstream's
/// `add_assign_mixed` of the identity is a no-op; the library leaves such entries out of its bucket lists (msm.hip.h msm_base_is_inf)
fn msm_inf<E: HipCurve>(a: &Arrays, group: usize, out: &mut String) {
    let c = E::CURVE_ID;
    let pre = format!("x_msm_inf.c{}_g{}_", c, group);
    let sname = format!("{}scalars", pre);
    let sc: Vec<<E::Fr as PrimeField>::BigInt> =
        a[&sname].1.chunks(a[&sname].0[1]).map(|l| { let mut b = <E::Fr as PrimeField>::BigInt::default(); b.as_mut().copy_from_slice(l); b }).collect();
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    if group == 1 {
        let r = VariableBaseMSM::multi_scalar_mul(&g1s::<E>(a, &format!("{}bases", pre), &format!("{}inf", pre)), &sc).into_affine();
        E::push_g1(&r, &mut xy, &mut inf);
        if r.is_zero() { xy.iter_mut().for_each(|w| *w = 0); }
    } else {
        let r = VariableBaseMSM::multi_scalar_mul(&g2s::<E>(a, &format!("{}bases", pre), &format!("{}inf", pre)), &sc).into_affine();
        E::push_g2(&r, &mut xy, &mut inf);
        if r.is_zero() { xy.iter_mut().for_each(|w| *w = 0); }
    }
    emit(out, &format!("{}result_xy", pre), "uint64", &[xy.len()], &xy);
    emit(out, &format!("{}result_inf", pre), "uint8", &[1], &[inf[0] as u64]);
}

fn fft<F: PrimeField + ark_ff::FftField>(a: &Arrays, fid: usize, out: &mut String) {
    for (name, arr) in a.iter().filter(|(k, _)| k.starts_with(&format!("fft.f{}_n", fid)) && k.ends_with("_in")) {
        let x: Vec<F> = fr_vec(arr);
        let dom = Radix2EvaluationDomain::<F>::new(x.len()).unwrap();
        let stem = &name[..name.len() - 3];
        for (inv, coset) in [(0, 0), (0, 1), (1, 0), (1, 1)].iter() {
            let mut v = x.clone();
            match (inv, coset) {
                (0, 0) => dom.fft_in_place(&mut v), (0, 1) => dom.coset_fft_in_place(&mut v),
                (1, 0) => dom.ifft_in_place(&mut v), _ => dom.coset_ifft_in_place(&mut v),
            }
            emit(out, &format!("{}_i{}c{}", stem, inv, coset), "uint64", &arr.0, &fr_limbs(&v));
        }
    }
}

fn fqk_limbs<E: HipCurve>(g: &E::Fqk) -> Vec<u64> {
    // GT in tower order c0, c1 over Fq2 / Fq3, base-field coefficients in order: exactly the element's `to_base_prime_field_elements`
    let mut o = Vec::new();
    for c in g.to_base_prime_field_elements() { marshal::push_fp(&c, &mut o); }
    o
}
fn pairing<E: HipCurve>(a: &Arrays, out: &mut String) {
    let c = E::CURVE_ID;
    let p = E::g1_from(&a[&format!("pairing.c{}_p", c)].1, false);
    let q = E::g2_from(&a[&format!("pairing.c{}_q", c)].1, false);
    let gt = E::pairing(p, q);
    let l = fqk_limbs::<E>(&gt);
    emit(out, &format!("pairing.c{}_gt", c), "uint64", &[l.len()], &l);
}

/// the golden constraint system, replayed: variables carry the golden assignment, every row is enforced as it stands
struct Replay<F: PrimeField> { rows: [Vec<Vec<(F, usize)>>; 3], z: Vec<F>, num_inputs: usize }
impl<F: PrimeField> ConstraintSynthesizer<F> for Replay<F> {
    fn generate_constraints(self, cs: ConstraintSystemRef<F>) -> Result<(), SynthesisError> {
        let mut vars = vec![Variable::One];
        for (i, v) in self.z.iter().enumerate().skip(1) {
            let v = *v;
            vars.push(if i < self.num_inputs { cs.new_input_variable(|| Ok(v))? } else { cs.new_witness_variable(|| Ok(v))? });
        }
        let lc_of = |row: &Vec<(F, usize)>| -> LinearCombination<F> { let mut l = lc!(); for (c, j) in row { l = l + (*c, vars[*j]); } l };
        for j in 0..self.rows[0].len() { cs.enforce_constraint(lc_of(&self.rows[0][j]), lc_of(&self.rows[1][j]), lc_of(&self.rows[2][j]))?; }
        Ok(())
    }
}
fn csr<F: PrimeField>(a: &Arrays, pre: &str, m: &str) -> Vec<Vec<(F, usize)>> {
    let rp = &a[&format!("{}rp_{}", pre, m)].1;
    let col = &a[&format!("{}col_{}", pre, m)].1;
    let cf: Vec<F> = fr_vec(&a[&format!("{}coeff_{}", pre, m)]);
    (0..rp.len() - 1).map(|j| (rp[j] as usize..rp[j + 1] as usize).map(|k| (cf[k], col[k] as usize)).collect()).collect()
}
fn groth16<E: HipCurve>(a: &Arrays, out: &mut String) {
    let pre = format!("groth16.c{}_", E::CURVE_ID);
    let z: Vec<E::Fr> = fr_vec(&a[&format!("{}z", pre)]);
    let ni = a[&format!("{}num_inputs", pre)].1[0] as usize;
    let one1 = |k: &str| E::g1_from(&a[&format!("{}{}", pre, k)].1, false);
    let one2 = |k: &str| E::g2_from(&a[&format!("{}{}", pre, k)].1, false);
    let q1 = |k: &str| g1s::<E>(a, &format!("{}{}", pre, k), &format!("{}{}_inf", pre, k));
    let vk = VerifyingKey::<E> { alpha_g1: one1("alpha_g1"), beta_g2: one2("beta_g2"), gamma_g2: one2("gamma_g2"), delta_g2: one2("delta_g2"),
                                 gamma_abc_g1: q1("gamma_abc_g1") };
    let pk = ProvingKey::<E> { vk, beta_g1: one1("beta_g1"), delta_g1: one1("delta_g1"), a_query: q1("a_query"), b_g1_query: q1("b_g1_query"),
                               b_g2_query: g2s::<E>(a, &format!("{}b_g2_query", pre), &format!("{}b_g2_query_inf", pre)),
                               h_query: q1("h_query"), l_query: q1("l_query") };
    {   // the witness map alone (groth16.c*_h): same synthesis settings as `create_proof`
        let again = Replay::<E::Fr> { rows: [csr(a, &pre, "a"), csr(a, &pre, "b"), csr(a, &pre, "c")], z: z.clone(), num_inputs: ni };
        let cs = ConstraintSystem::<E::Fr>::new_ref();
        cs.set_optimization_goal(OptimizationGoal::Constraints);
        again.generate_constraints(cs.clone()).unwrap();
        cs.finalize();
        let h = LibsnarkReduction::witness_map::<E::Fr, GeneralEvaluationDomain<E::Fr>>(cs).unwrap();
        emit(out, &format!("{}h", pre), "uint64", &[h.len(), (E::Fr::size_in_bits() + 63) / 64], &fr_limbs(&h));
    }
    let circuit = Replay::<E::Fr> { rows: [csr(a, &pre, "a"), csr(a, &pre, "b"), csr(a, &pre, "c")], z, num_inputs: ni };
    let r: E::Fr = marshal::fp_from_limbs(&a[&format!("{}r", pre)].1);
    let s: E::Fr = marshal::fp_from_limbs(&a[&format!("{}s", pre)].1);
    let proof = create_proof_with_reduction::<E, _, LibsnarkReduction>(circuit, &pk, r, s).unwrap();
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    E::push_g1(&proof.a, &mut xy, &mut inf); E::push_g2(&proof.b, &mut xy, &mut inf); E::push_g1(&proof.c, &mut xy, &mut inf);
    emit(out, &format!("{}proof", pre), "uint64", &[xy.len()], &xy);
}

fn wire<E: HipCurve>(a: &Arrays, out: &mut String) {
    let c = E::CURVE_ID;
    let p1 = g1s::<E>(a, &format!("wire.c{}_g1_xy", c), &format!("wire.c{}_g1_inf", c));
    let p2 = g2s::<E>(a, &format!("wire.c{}_g2_xy", c), &format!("wire.c{}_g2_inf", c));
    let bytes = |v: &Vec<u8>| -> Vec<u64> { v.iter().map(|b| *b as u64).collect() };
    for comp in 0..2 {
        let (mut b1, mut b2) = (Vec::new(), Vec::new());
        for p in &p1 { if comp == 1 { p.serialize(&mut b1).unwrap() } else { p.serialize_uncompressed(&mut b1).unwrap() } }
        for p in &p2 { if comp == 1 { p.serialize(&mut b2).unwrap() } else { p.serialize_uncompressed(&mut b2).unwrap() } }
        emit(out, &format!("wire.c{}_g1_ser{}", c, comp), "uint8", &[b1.len()], &bytes(&b1));
        emit(out, &format!("wire.c{}_g2_ser{}", c, comp), "uint8", &[b2.len()], &bytes(&b2));
        // the proof-shaped triple and the verifying-key-shaped tuple tests/golden/gen_golden.py builds from the same six points
        let proof = Proof::<E> { a: p1[0], b: p2[3], c: p1[4] };
        let vk = VerifyingKey::<E> { alpha_g1: p1[3], beta_g2: p2[0], gamma_g2: p2[1], delta_g2: p2[4], gamma_abc_g1: vec![p1[0], p1[1], p1[5]] };
        let (mut bp, mut bv) = (Vec::new(), Vec::new());
        if comp == 1 { proof.serialize(&mut bp).unwrap(); vk.serialize(&mut bv).unwrap(); }
        else { proof.serialize_uncompressed(&mut bp).unwrap(); vk.serialize_uncompressed(&mut bv).unwrap(); }
        emit(out, &format!("wire.c{}_proof_ser{}", c, comp), "uint8", &[bp.len()], &bytes(&bp));
        emit(out, &format!("wire.c{}_vk_ser{}", c, comp), "uint8", &[bv.len()], &bytes(&bv));
    }
}

// ---------------------------------------------------------------------------------------------- round 4: what is only recalled

/// all four transforms of `MixedRadixEvaluationDomain` on the sizes tools/kat_extra.py exported for this field
fn mixed<F: PrimeField + FftField>(a: &Arrays, fid: usize, out: &mut String) {
    for (name, arr) in a.iter().filter(|(k, _)| k.starts_with(&format!("x_mixed.f{}_n", fid)) && k.ends_with("_in")) {
        let x: Vec<F> = fr_vec(arr);
        let dom = MixedRadixEvaluationDomain::<F>::new(x.len()).expect("a 2^a q^b size within the field's adicities");
        assert_eq!(dom.size(), x.len(), "the mixed-radix domain must have exactly the exported size");
        let stem = &name[..name.len() - 3];
        for (inv, coset) in [(0, 0), (0, 1), (1, 0), (1, 1)].iter() {
            let mut v = x.clone();
            match (inv, coset) {
                (0, 0) => dom.fft_in_place(&mut v), (0, 1) => dom.coset_fft_in_place(&mut v),
                (1, 0) => dom.ifft_in_place(&mut v), _ => dom.coset_ifft_in_place(&mut v),
            }
            emit(out, &format!("{}_i{}c{}", stem, inv, coset), "uint64", &arr.0, &fr_limbs(&v));
        }
    }
}

/// `LibsnarkReduction::witness_map` over `GeneralEvaluationDomain` for a circuit past the 2-adicity of E::Fr: the domain upstream
/// picks (size) and h on it.  Same synthesis settings as `create_proof` (optimisation goal, `finalize`).
fn wm_mixed<E: HipCurve>(a: &Arrays, out: &mut String) {
    let pre = "x_wm.";
    let z: Vec<E::Fr> = fr_vec(&a[&format!("{}z", pre)]);
    let ni = a[&format!("{}num_inputs", pre)].1[0] as usize;
    let circuit = Replay::<E::Fr> { rows: [csr(a, pre, "a"), csr(a, pre, "b"), csr(a, pre, "c")], z, num_inputs: ni };
    let cs = ConstraintSystem::<E::Fr>::new_ref();
    cs.set_optimization_goal(OptimizationGoal::Constraints);
    circuit.generate_constraints(cs.clone()).unwrap();
    assert!(cs.is_satisfied().unwrap());
    cs.finalize();
    let dom = GeneralEvaluationDomain::<E::Fr>::new(cs.num_constraints() + cs.num_instance_variables()).unwrap();
    emit(out, "x_wm.domain_size", "uint64", &[1], &[dom.size() as u64]);
    let h = LibsnarkReduction::witness_map::<E::Fr, GeneralEvaluationDomain<E::Fr>>(cs).unwrap();
    emit(out, "x_wm.h", "uint64", &[h.len(), (E::Fr::size_in_bits() + 63) / 64], &fr_limbs(&h));
}

fn scalars_fr<E: HipCurve>(a: &Arrays, name: &str) -> Vec<E::Fr> {   // canonical limbs -> field elements
    a[name].1.chunks(a[name].0[1]).map(|l| {
        let mut b = <E::Fr as PrimeField>::BigInt::default();
        b.as_mut().copy_from_slice(l);
        E::Fr::from_repr(b).expect("reduced scalar")
    }).collect()
}
/// `FixedBaseMSM` exactly as ark-groth16's generator drives it: window from `get_mul_window_size`, table, batch, normalisation
fn fixed_base<E: HipCurve>(a: &Arrays, out: &mut String) {
    let c = E::CURVE_ID;
    let bits = E::Fr::size_in_bits();
    {
        let base = E::g1_from(&a[&format!("x_fixed.c{}_g1_base", c)].1, false);
        let sc = scalars_fr::<E>(a, &format!("x_fixed.c{}_g1_scalars", c));
        let w = FixedBaseMSM::get_mul_window_size(sc.len());
        let table = FixedBaseMSM::get_window_table::<E::G1Projective>(bits, w, base.into_projective());
        let res = FixedBaseMSM::multi_scalar_mul::<E::G1Projective>(bits, w, &table, &sc);
        let aff = E::G1Projective::batch_normalization_into_affine(&res);
        let (mut xy, mut inf) = (Vec::new(), Vec::new());
        for p in &aff { let k = xy.len(); E::push_g1(p, &mut xy, &mut inf); if p.is_zero() { xy[k..].iter_mut().for_each(|w| *w = 0); } }
        emit(out, &format!("x_fixed.c{}_g1_out_xy", c), "uint64", &[aff.len(), 2 * E::FQ_LIMBS], &xy);
        emit(out, &format!("x_fixed.c{}_g1_out_inf", c), "uint8", &[aff.len()], &inf.iter().map(|b| *b as u64).collect::<Vec<_>>());
    }
    {
        let base = E::g2_from(&a[&format!("x_fixed.c{}_g2_base", c)].1, false);
        let sc = scalars_fr::<E>(a, &format!("x_fixed.c{}_g2_scalars", c));
        let w = FixedBaseMSM::get_mul_window_size(sc.len());
        let table = FixedBaseMSM::get_window_table::<E::G2Projective>(bits, w, base.into_projective());
        let res = FixedBaseMSM::multi_scalar_mul::<E::G2Projective>(bits, w, &table, &sc);
        let aff = E::G2Projective::batch_normalization_into_affine(&res);
        let (mut xy, mut inf) = (Vec::new(), Vec::new());
        for p in &aff { let k = xy.len(); E::push_g2(p, &mut xy, &mut inf); if p.is_zero() { xy[k..].iter_mut().for_each(|w| *w = 0); } }
        emit(out, &format!("x_fixed.c{}_g2_out_xy", c), "uint64", &[aff.len(), 2 * E::G2_DEG * E::FQ_LIMBS], &xy);
        emit(out, &format!("x_fixed.c{}_g2_out_inf", c), "uint8", &[aff.len()], &inf.iter().map(|b| *b as u64).collect::<Vec<_>>());
    }
}

/// An RNG that replays given u64 words: `Fp::rand` fills the element's `BigInteger` limb by limb from `next_u64` (Montgomery image,
/// then a range check), so feeding it the Montgomery limbs of tau makes `sample_element_outside_domain` return exactly tau.
struct Replayed { words: Vec<u64>, at: usize }
impl ark_std::rand::RngCore for Replayed {
    fn next_u32(&mut self) -> u32 { self.next_u64() as u32 }
    fn next_u64(&mut self) -> u64 { let w = self.words[self.at % self.words.len()]; self.at += 1; w }
    fn fill_bytes(&mut self, dest: &mut [u8]) { for ch in dest.chunks_mut(8) { let w = self.next_u64().to_le_bytes(); ch.copy_from_slice(&w[..ch.len()]); } }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), ark_std::rand::Error> { self.fill_bytes(dest); Ok(()) }
}
/// ark-groth16 `generate_parameters` (the body of `circuit_specific_setup`, /root/reference src/ec_cycle_pcd/mod.rs:69,78) with the
/// golden toxic waste and the oracle's generators: every query of the key goes back out under the golden names.
/// (Signature of the era the reference tracks: `generate_parameters(circuit, alpha, beta, gamma, delta, g1_generator, g2_generator, rng)`;
/// tau is the first `Fr::rand(rng)` outside the domain.)
fn setup<E: HipCurve>(a: &Arrays, out: &mut String) { setup_at::<E>(a, &format!("groth16.c{}_", E::CURVE_ID), true, out) }
/// round 5: the same over a system with variables that no row of A / B mentions (tools/kat_extra.py x_setup_inf): which entries of the
/// a / b queries upstream leaves as the identity, flag and all -- only those three queries are written back
fn setup_inf<E: HipCurve>(a: &Arrays, out: &mut String) { setup_at::<E>(a, "x_setup_inf.", false, out) }
fn setup_at<E: HipCurve>(a: &Arrays, pre: &str, whole_key: bool, out: &mut String) {
    let pre = pre.to_string();
    let z: Vec<E::Fr> = fr_vec(&a[&format!("{}z", pre)]);
    let ni = a[&format!("{}num_inputs", pre)].1[0] as usize;
    let tox: Vec<E::Fr> = fr_vec(&a[&format!("{}toxic", pre)]);   // alpha, beta, gamma, delta, tau
    let l = (E::Fr::size_in_bits() + 63) / 64;
    let mut rng = Replayed { words: a[&format!("{}toxic", pre)].1[4 * l..5 * l].to_vec(), at: 0 };
    let g1 = E::g1_from(&a[&format!("x_gens.c{}_g1", E::CURVE_ID)].1, false).into_projective();
    let g2 = E::g2_from(&a[&format!("x_gens.c{}_g2", E::CURVE_ID)].1, false).into_projective();
    let circuit = Replay::<E::Fr> { rows: [csr(a, &pre, "a"), csr(a, &pre, "b"), csr(a, &pre, "c")], z, num_inputs: ni };
    let pk: ProvingKey<E> = generate_parameters::<E, _, _>(circuit, tox[0], tox[1], tox[2], tox[3], g1, g2, &mut rng).unwrap();
    let mut put1 = |name: &str, pts: &[E::G1Affine], flags: bool| {
        let (mut xy, mut inf) = (Vec::new(), Vec::new());
        for p in pts { let k = xy.len(); E::push_g1(p, &mut xy, &mut inf); if p.is_zero() { xy[k..].iter_mut().for_each(|w| *w = 0); } }
        let shape: Vec<usize> = if flags { vec![pts.len(), 2 * E::FQ_LIMBS] } else { vec![2 * E::FQ_LIMBS] };
        emit(out, &format!("{}{}", pre, name), "uint64", &shape, &xy);
        if flags { emit(out, &format!("{}{}_inf", pre, name), "uint8", &[pts.len()], &inf.iter().map(|b| *b as u64).collect::<Vec<_>>()); }
    };
    if whole_key {
        put1("alpha_g1", core::slice::from_ref(&pk.vk.alpha_g1), false);
        put1("beta_g1", core::slice::from_ref(&pk.beta_g1), false);
        put1("delta_g1", core::slice::from_ref(&pk.delta_g1), false);
    }
    put1("a_query", &pk.a_query, true);
    put1("b_g1_query", &pk.b_g1_query, true);
    if whole_key {
        put1("h_query", &pk.h_query, true);
        put1("l_query", &pk.l_query, true);
        put1("gamma_abc_g1", &pk.vk.gamma_abc_g1, true);
    }
    let mut put2 = |name: &str, pts: &[E::G2Affine], flags: bool| {
        let (mut xy, mut inf) = (Vec::new(), Vec::new());
        for p in pts { let k = xy.len(); E::push_g2(p, &mut xy, &mut inf); if p.is_zero() { xy[k..].iter_mut().for_each(|w| *w = 0); } }
        let w = 2 * E::G2_DEG * E::FQ_LIMBS;
        let shape: Vec<usize> = if flags { vec![pts.len(), w] } else { vec![w] };
        emit(out, &format!("{}{}", pre, name), "uint64", &shape, &xy);
        if flags { emit(out, &format!("{}{}_inf", pre, name), "uint8", &[pts.len()], &inf.iter().map(|b| *b as u64).collect::<Vec<_>>()); }
    };
    if whole_key {
        put2("beta_g2", core::slice::from_ref(&pk.vk.beta_g2), false);
        put2("gamma_g2", core::slice::from_ref(&pk.vk.gamma_g2), false);
        put2("delta_g2", core::slice::from_ref(&pk.vk.delta_g2), false);
    }
    put2("b_g2_query", &pk.b_g2_query, true);
}

/// the constants themselves: per field the multiplicative generator (the coset shift of every coset transform), the 2-adic root of unity,
/// the 2-adicity and -- where upstream defines them -- the small-subgroup base and its adicity; per curve the group generators
fn consts_field<F: PrimeField + FftField>(fid: usize, out: &mut String) {
    emit(out, &format!("x_consts.f{}_generator", fid), "uint64", &[(F::size_in_bits() + 63) / 64], &fr_limbs(&[F::multiplicative_generator()]));
    emit(out, &format!("x_consts.f{}_two_adic_root", fid), "uint64", &[(F::size_in_bits() + 63) / 64], &fr_limbs(&[F::two_adic_root_of_unity()]));
    emit(out, &format!("x_consts.f{}_two_adicity", fid), "uint64", &[1], &[<F::FftParams as FftParameters>::TWO_ADICITY as u64]);
    if let Some(b) = <F::FftParams as FftParameters>::SMALL_SUBGROUP_BASE {
        emit(out, &format!("x_consts.f{}_small_subgroup_base", fid), "uint64", &[1], &[b as u64]);
        emit(out, &format!("x_consts.f{}_small_subgroup_base_adicity", fid), "uint64", &[1],
             &[<F::FftParams as FftParameters>::SMALL_SUBGROUP_BASE_ADICITY.unwrap() as u64]);
    }
}
fn consts_curve<E: HipCurve>(out: &mut String) {
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    E::push_g1(&E::G1Affine::prime_subgroup_generator(), &mut xy, &mut inf);
    emit(out, &format!("x_consts.c{}_g1_generator", E::CURVE_ID), "uint64", &[xy.len()], &xy);
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    E::push_g2(&E::G2Affine::prime_subgroup_generator(), &mut xy, &mut inf);
    emit(out, &format!("x_consts.c{}_g2_generator", E::CURVE_ID), "uint64", &[xy.len()], &xy);
}

#[test]
fn kat() {
    let a = load();
    let mut out = String::new();
    msm::<ark_mnt4_298::MNT4_298>(&a, &mut out); msm::<ark_mnt6_298::MNT6_298>(&a, &mut out);
    msm::<ark_mnt4_753::MNT4_753>(&a, &mut out); msm::<ark_mnt6_753::MNT6_753>(&a, &mut out);
    fft::<ark_mnt4_298::Fq>(&a, 0, &mut out); fft::<ark_mnt4_298::Fr>(&a, 1, &mut out);
    fft::<ark_mnt4_753::Fq>(&a, 2, &mut out); fft::<ark_mnt4_753::Fr>(&a, 3, &mut out);
    pairing::<ark_mnt4_298::MNT4_298>(&a, &mut out); pairing::<ark_mnt6_298::MNT6_298>(&a, &mut out);
    pairing::<ark_mnt4_753::MNT4_753>(&a, &mut out); pairing::<ark_mnt6_753::MNT6_753>(&a, &mut out);
    groth16::<ark_mnt4_298::MNT4_298>(&a, &mut out); groth16::<ark_mnt6_298::MNT6_298>(&a, &mut out);
    wire::<ark_mnt4_298::MNT4_298>(&a, &mut out); wire::<ark_mnt6_298::MNT6_298>(&a, &mut out);
    wire::<ark_mnt4_753::MNT4_753>(&a, &mut out); wire::<ark_mnt6_753::MNT6_753>(&a, &mut out);
    // round 4
    mixed::<ark_mnt4_298::Fq>(&a, 0, &mut out); mixed::<ark_mnt4_753::Fq>(&a, 2, &mut out);
    wm_mixed::<ark_mnt6_753::MNT6_753>(&a, &mut out);   // Fr of MNT6-753 = field 2 (2-adicity 15)
    fixed_base::<ark_mnt4_298::MNT4_298>(&a, &mut out); fixed_base::<ark_mnt6_298::MNT6_298>(&a, &mut out);
    fixed_base::<ark_mnt4_753::MNT4_753>(&a, &mut out); fixed_base::<ark_mnt6_753::MNT6_753>(&a, &mut out);
    setup::<ark_mnt4_298::MNT4_298>(&a, &mut out); setup::<ark_mnt6_298::MNT6_298>(&a, &mut out);
    consts_field::<ark_mnt4_298::Fq>(0, &mut out); consts_field::<ark_mnt4_298::Fr>(1, &mut out);
    consts_field::<ark_mnt4_753::Fq>(2, &mut out); consts_field::<ark_mnt4_753::Fr>(3, &mut out);
    consts_curve::<ark_mnt4_298::MNT4_298>(&mut out); consts_curve::<ark_mnt6_298::MNT6_298>(&mut out);
    consts_curve::<ark_mnt4_753::MNT4_753>(&mut out); consts_curve::<ark_mnt6_753::MNT6_753>(&mut out);
    // round 5
    fields::<ark_mnt4_298::Fq>(&a, 0, &mut out); fields::<ark_mnt4_298::Fr>(&a, 1, &mut out);
    fields::<ark_mnt4_753::Fq>(&a, 2, &mut out); fields::<ark_mnt4_753::Fr>(&a, 3, &mut out);
    setup_inf::<ark_mnt6_298::MNT6_298>(&a, &mut out);
    msm_inf::<ark_mnt4_298::MNT4_298>(&a, 1, &mut out); msm_inf::<ark_mnt4_298::MNT4_298>(&a, 2, &mut out);
    msm_inf::<ark_mnt6_753::MNT6_753>(&a, 2, &mut out);
    // what was written is what the manifest says, both ways
    let written: std::collections::BTreeSet<String> = out.lines().map(|l| family(l.split(' ').next().unwrap())).collect();
    for f in &written { assert!(EMITS.contains(&f.as_str()), "line family {} is not in EMITS", f); }
    for f in EMITS { assert!(written.contains(*f), "EMITS names {} but no such line was written", f); }
    std::fs::write(concat!(env!("CARGO_MANIFEST_DIR"), "/tests/kat_outputs.txt"), &out).unwrap();
    println!("wrote {} lines to rust/tests/kat_outputs.txt: now run `python tools/check_kat.py`", out.lines().count());
}
