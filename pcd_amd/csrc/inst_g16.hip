// One object per curve (compile with -DPCD_CURVE_IDX=0..3): Groth16 proof assembly (K5 of SURVEY.md
// section 8) -- ark-groth16 `create_proof` after the five MSMs:
//   A = alpha + a_query[0] + r*delta + M_a          B = beta + b_query[0] + s*delta + M_b   (G1 and G2)
//   C = s*A + r*B_1 - r s*delta + M_l + M_h  =  s*(alpha + a_0 + M_a) + r*(beta_1 + b_0 + M_b1) + r s*delta + M_l + M_h
// r*delta, s*delta and -rs*delta are ordinary (base, scalar) pairs appended to the MSMs.
#include "common.h"

namespace pcd {

// (the 298-bit curves use the compact field variant here: these kernels are single-lane and latency-bound, and a
//  double-and-add loop body with inlined 242-mad products does not fit the instruction cache)
#if PCD_CURVE_IDX == 0
typedef G1_MNT4_298_C GA; typedef G2_MNT4_298_C GB;
#elif PCD_CURVE_IDX == 1
typedef G1_MNT6_298_C GA; typedef G2_MNT6_298_C GB;
#elif PCD_CURVE_IDX == 2
typedef G1_MNT4_753 GA; typedef G2_MNT4_753 GB;
#elif PCD_CURVE_IDX == 3
typedef G1_MNT6_753 GA; typedef G2_MNT6_753 GB;
#else
#error "PCD_CURVE_IDX must be 0..3"
#endif

namespace {

typedef typename GA::F F1;
typedef typename GB::F F2;
typedef Fp<typename GA::FR, false> FR;
// device-internal word counts, and the C-ABI ones (suffix A)
constexpr int J1 = Jac<F1>::WORDS, J2 = Jac<F2>::WORDS, A1 = Aff<F1>::WORDS, A2 = Aff<F2>::WORDS;
constexpr int A1A = Aff<F1>::ABI_WORDS, A2A = Aff<F2>::ABI_WORDS, SWA = FR::ABI_WORDS;

// singles: alpha_g1, beta_g1, delta_g1, a_0, b1_0 (G1 affine) then beta_g2, delta_g2, b2_0 (G2 affine)
__global__ void __launch_bounds__(64) g16_singles_in(const uint32_t* __restrict__ abi, uint32_t* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x >= 8) return;
  int i = threadIdx.x;
  if (i < 5) Aff<F1>::from_abi(abi + i * A1A).store(out + i * A1);
  else Aff<F2>::from_abi(abi + 5 * A1A + (i - 5) * A2A).store(out + 5 * A1 + (i - 5) * A2);
}

// Scalars the MSMs consume besides the assignment (canonical words): out = [r, s, -(r s)];  rs = ABI Montgomery (r, s)
__global__ void __launch_bounds__(64) g16_prepare_scalars(const uint32_t* __restrict__ rs, uint32_t* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  FR r = FR::from_abi(rs), s = FR::from_abi(rs + SWA);
  r.to_canonical_words(out);
  s.to_canonical_words(out + SWA);
  (r * s).neg().to_canonical_words(out + 2 * SWA);
}

// The fixed-base terms r*delta, s*delta, -rs*delta ride inside the MSMs (delta is appended to the a / b / l
// queries at key upload), so only the two variable-base products s*A and r*B_1 remain:
//   A = alpha + a_0 + M_a'      B_1 = beta_1 + b_0 + M_b1'     (M' include the delta terms)
//   C = s*A + r*B_1 + M_l' + M_h
__global__ void __launch_bounds__(64) g16_scalar_muls(const uint32_t* __restrict__ msm_g1, const uint32_t* __restrict__ singles,
                                                      const uint32_t* __restrict__ rs, uint32_t* __restrict__ scratch) {
  if (threadIdx.x != 0) return;
  typedef EC<GA> E1;
  uint32_t k[SWA];
  if (blockIdx.x == 0) {
    FR::from_abi(rs + SWA).to_canonical_words(k);  // s
    Jac<F1> A = E1::madd(E1::madd(Jac<F1>::load(msm_g1 + 2 * J1), Aff<F1>::load(singles + 3 * A1)), Aff<F1>::load(singles));
    A.store(scratch);
    E1::mul(A, k, SWA).store(scratch + J1);
  } else {
    FR::from_abi(rs).to_canonical_words(k);  // r
    Jac<F1> B = E1::madd(E1::madd(Jac<F1>::load(msm_g1 + 3 * J1), Aff<F1>::load(singles + 4 * A1)), Aff<F1>::load(singles + A1));
    E1::mul(B, k, SWA).store(scratch + 2 * J1);
  }
}

__global__ void __launch_bounds__(64) g16_finish(const uint32_t* __restrict__ msm_g1, const uint32_t* __restrict__ msm_g2,
                                                 const uint32_t* __restrict__ singles, const uint32_t* __restrict__ scratch,
                                                 uint32_t* __restrict__ proof_abi) {
  if (threadIdx.x != 0) return;
  typedef EC<GA> E1;
  typedef EC<GB> E2;
  if (blockIdx.x == 0) {
    E1::to_affine(Jac<F1>::load(scratch)).to_abi(proof_abi);
  } else if (blockIdx.x == 1) {  // B (G2) = beta_2 + b2_0 + M_b2'
    Jac<F2> t = E2::madd(E2::madd(Jac<F2>::load(msm_g2), Aff<F2>::load(singles + 5 * A1 + 2 * A2)), Aff<F2>::load(singles + 5 * A1));
    E2::to_affine(t).to_abi(proof_abi + A1A);
  } else {  // C = s A + r B_1 + M_l' + M_h
    Jac<F1> t = E1::add(Jac<F1>::load(scratch + J1), Jac<F1>::load(scratch + 2 * J1));
    t = E1::add(t, Jac<F1>::load(msm_g1 + J1));
    t = E1::add(t, Jac<F1>::load(msm_g1));
    E1::to_affine(t).to_abi(proof_abi + A1A + A2A);
  }
}

hipError_t singles_in(hipStream_t st, const uint32_t* abi, uint32_t* out) {
  hipLaunchKernelGGL(g16_singles_in, dim3(1), dim3(64), 0, st, abi, out);
  return hipGetLastError();
}
hipError_t prepare_scalars(hipStream_t st, const uint32_t* rs_dev, uint32_t* out3) {
  hipLaunchKernelGGL(g16_prepare_scalars, dim3(1), dim3(64), 0, st, rs_dev, out3);
  return hipGetLastError();
}

hipError_t assemble(hipStream_t st, const uint32_t* msm_g1, const uint32_t* msm_g2, const uint32_t* singles, const uint32_t* rs_dev,
                    uint32_t* scratch, uint32_t* proof_out) {
  hipLaunchKernelGGL(g16_scalar_muls, dim3(2), dim3(64), 0, st, msm_g1, singles, rs_dev, scratch);
  hipLaunchKernelGGL(g16_finish, dim3(3), dim3(64), 0, st, msm_g1, msm_g2, singles, scratch, proof_out);
  return hipGetLastError();
}

}  // namespace

#define PCD_CAT_(a, b) a##b
#define PCD_CAT(a, b) PCD_CAT_(a, b)
const CurveEntry* PCD_CAT(pcd_curve_entry_, PCD_CURVE_IDX)() {
  static const CurveEntry e = {prepare_scalars, singles_in, (size_t)(5 * A1A + 3 * A2A) * 4, (size_t)(5 * A1 + 3 * A2) * 4,
                               (size_t)(2 * A1A + A2A) * 4, assemble, (size_t)(3 * J1) * 4};
  return &e;
}

}  // namespace pcd
