// One object per curve (compile with -DPCD_CURVE_IDX=0..3): Groth16 proof assembly (K5 of SURVEY.md
// section 8) -- ark-groth16 `create_proof` after the five MSMs:
//   A = alpha + a_query[0] + r*delta + M_a          B = beta + b_query[0] + s*delta + M_b   (G1 and G2)
//   C = s*A + r*B_1 - r s*delta + M_l + M_h  =  s*(alpha + a_0 + M_a) + r*(beta_1 + b_0 + M_b1) + r s*delta + M_l + M_h
// r*delta, s*delta and -rs*delta are ordinary (base, scalar) pairs appended to the MSMs.
#include "common.h"

namespace pcd {

#if PCD_CURVE_IDX == 0
typedef G1_MNT4_298 GA; typedef G2_MNT4_298 GB;
#elif PCD_CURVE_IDX == 1
typedef G1_MNT6_298 GA; typedef G2_MNT6_298 GB;
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
typedef Fp<typename GA::FR> FR;
constexpr int J1 = Jac<F1>::WORDS, J2 = Jac<F2>::WORDS, A1 = Aff<F1>::WORDS, A2 = Aff<F2>::WORDS, SW = FR::WORDS;

// Scalars the MSMs consume besides the assignment (canonical form): out = [r, s, -(r s)]
__global__ void __launch_bounds__(64) g16_prepare_scalars(const uint32_t* __restrict__ rs, uint32_t* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  FR r = FR::load(rs), s = FR::load(rs + SW);
  r.from_mont().store(out);
  s.from_mont().store(out + SW);
  (r * s).neg().from_mont().store(out + 2 * SW);
}

// The fixed-base terms r*delta, s*delta, -rs*delta ride inside the MSMs (delta is appended to the a / b / l
// queries at key upload), so only the two variable-base products s*A and r*B_1 remain:
//   A = alpha + a_0 + M_a'      B_1 = beta_1 + b_0 + M_b1'     (M' include the delta terms)
//   C = s*A + r*B_1 + M_l' + M_h
__global__ void __launch_bounds__(64) g16_scalar_muls(const uint32_t* __restrict__ msm_g1, const uint32_t* __restrict__ singles,
                                                      const uint32_t* __restrict__ rs, uint32_t* __restrict__ scratch) {
  if (threadIdx.x != 0) return;
  typedef EC<GA> E1;
  if (blockIdx.x == 0) {
    FR k = FR::load(rs + SW).from_mont();  // s
    Jac<F1> A = E1::madd(E1::madd(Jac<F1>::load(msm_g1 + 2 * J1), Aff<F1>::load(singles + 3 * A1)), Aff<F1>::load(singles));
    A.store(scratch);
    E1::mul(A, k.v, SW).store(scratch + J1);
  } else {
    FR k = FR::load(rs).from_mont();  // r
    Jac<F1> B = E1::madd(E1::madd(Jac<F1>::load(msm_g1 + 3 * J1), Aff<F1>::load(singles + 4 * A1)), Aff<F1>::load(singles + A1));
    E1::mul(B, k.v, SW).store(scratch + 2 * J1);
  }
}

__global__ void __launch_bounds__(64) g16_finish(const uint32_t* __restrict__ msm_g1, const uint32_t* __restrict__ msm_g2,
                                                 const uint32_t* __restrict__ singles, const uint32_t* __restrict__ scratch,
                                                 uint32_t* __restrict__ proof) {
  if (threadIdx.x != 0) return;
  typedef EC<GA> E1;
  typedef EC<GB> E2;
  if (blockIdx.x == 0) {
    E1::to_affine(Jac<F1>::load(scratch)).store(proof);
  } else if (blockIdx.x == 1) {  // B (G2) = beta_2 + b2_0 + M_b2'
    Jac<F2> t = E2::madd(E2::madd(Jac<F2>::load(msm_g2), Aff<F2>::load(singles + 5 * A1 + 2 * A2)), Aff<F2>::load(singles + 5 * A1));
    E2::to_affine(t).store(proof + A1);
  } else {  // C = s A + r B_1 + M_l' + M_h
    Jac<F1> t = E1::add(Jac<F1>::load(scratch + J1), Jac<F1>::load(scratch + 2 * J1));
    t = E1::add(t, Jac<F1>::load(msm_g1 + J1));
    t = E1::add(t, Jac<F1>::load(msm_g1));
    E1::to_affine(t).store(proof + A1 + A2);
  }
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
  static const CurveEntry e = {prepare_scalars, assemble, (size_t)(3 * J1) * 4};
  return &e;
}

}  // namespace pcd
