//! Rust values -> the plain buffers of include/pcdhip.h and back.  Field elements are copied as they sit in memory
//! (`Fp320` / `Fp768` = `BigInteger` limbs of the Montgomery residue: the C-ABI's encoding); `GroupAffine { x, y, infinity }`
//! is not `repr(C)`, so points are repacked as x || y plus a byte vector of infinity flags.
use ark_ff::PrimeField;
use ark_relations::r1cs::Matrix;

/// In-memory image of a prime-field element: the `BigInteger` limbs of its Montgomery representation.
/// ark-ff 0.2 / 0.3 lay `FpN<P>` out as `(BigIntegerN, PhantomData<P>)`: the limbs are the first (and only sized) field.
pub fn limbs_of<F: PrimeField>(x: &F) -> &[u64] {
    let n = (F::size_in_bits() + 63) / 64;
    debug_assert_eq!(core::mem::size_of::<F>(), 8 * n);
    unsafe { core::slice::from_raw_parts(x as *const F as *const u64, n) }
}
pub fn push_fp<F: PrimeField>(x: &F, out: &mut Vec<u64>) { out.extend_from_slice(limbs_of(x)); }
/// The inverse of `limbs_of`: a field element from Montgomery limbs the library returned (always reduced residues).
pub fn fp_from_limbs<F: PrimeField>(limbs: &[u64]) -> F {
    let n = (F::size_in_bits() + 63) / 64;
    assert_eq!(limbs.len(), n);
    let mut x = F::zero();
    unsafe { core::ptr::copy_nonoverlapping(limbs.as_ptr(), &mut x as *mut F as *mut u64, n) };
    x
}
/// Canonical (`into_repr()`) limbs, what `pcdhip_msm` takes for scalars.
pub fn push_repr<F: PrimeField>(x: &F, out: &mut Vec<u64>) { out.extend_from_slice(x.into_repr().as_ref()); }

/// `ConstraintMatrices::{a, b, c}` (rows of `(coefficient, column)`) -> CSR arrays in the C-ABI layout.
pub struct Csr { pub row_ptr: Vec<u64>, pub col: Vec<u32>, pub coeff: Vec<u64> }
impl Csr {
    pub fn from_matrix<F: PrimeField>(m: &Matrix<F>) -> Self {
        let nnz: usize = m.iter().map(|r| r.len()).sum();
        let mut row_ptr = Vec::with_capacity(m.len() + 1);
        let mut col = Vec::with_capacity(nnz);
        let mut coeff = Vec::with_capacity(nnz * ((F::size_in_bits() + 63) / 64));
        row_ptr.push(0u64);
        for row in m {
            for (c, j) in row {
                col.push(*j as u32);
                push_fp(c, &mut coeff);
            }
            row_ptr.push(col.len() as u64);
        }
        Csr { row_ptr, col, coeff }
    }
    pub fn view(&self) -> crate::ffi::pcdhip_csr {
        crate::ffi::pcdhip_csr { num_rows: (self.row_ptr.len() - 1) as u64, row_ptr: self.row_ptr.as_ptr(), col: self.col.as_ptr(), coeff: self.coeff.as_ptr() }
    }
}
