// One object per curve (compile with -DPCD_CURVE_IDX=0..3): Groth16 proof assembly (K5 of SURVEY.md
// section 8) -- ark-groth16 `create_proof` after the five MSMs:
//   A = alpha + a_query[0] + r*delta + M_a          B = beta + b_query[0] + s*delta + M_b   (G1 and G2)
//   C = s*A + r*B_1 - r s*delta + M_l + M_h  =  s*(alpha + a_0 + M_a) + r*(beta_1 + b_0 + M_b1) + r s*delta + M_l + M_h
// The six scalar multiplications are independent in this form and run in six workgroups.
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

__global__ void __launch_bounds__(64) g16_scalar_muls(const uint32_t* __restrict__ msm_g1, const uint32_t* __restrict__ singles,
                                                      const uint32_t* __restrict__ rs, uint32_t* __restrict__ scratch) {
  if (threadIdx.x != 0) return;
  FR r = FR::load(rs), s = FR::load(rs + SW);
  const uint32_t* alpha = singles;
  const uint32_t* beta1 = singles + A1;
  const uint32_t* delta1 = singles + 2 * A1;
  const uint32_t* a0 = singles + 3 * A1;
  const uint32_t* b10 = singles + 4 * A1;
  const uint32_t* delta2 = singles + 5 * A1 + A2;
  typedef EC<GA> E1;
  typedef EC<GB> E2;
  switch (blockIdx.x) {
    case 0: { FR k = r.from_mont(); E1::mul(Jac<F1>{Aff<F1>::load(delta1).x, Aff<F1>::load(delta1).y, F1::one()}, k.v, SW).store(scratch); } break;
    case 1: { FR k = s.from_mont(); E1::mul(Jac<F1>{Aff<F1>::load(delta1).x, Aff<F1>::load(delta1).y, F1::one()}, k.v, SW).store(scratch + J1); } break;
    case 2: {
      FR k = s.from_mont();
      Jac<F1> t = E1::madd(E1::madd(Jac<F1>::load(msm_g1 + 2 * J1), Aff<F1>::load(a0)), Aff<F1>::load(alpha));
      E1::mul(t, k.v, SW).store(scratch + 2 * J1);
    } break;
    case 3: {
      FR k = r.from_mont();
      Jac<F1> t = E1::madd(E1::madd(Jac<F1>::load(msm_g1 + 3 * J1), Aff<F1>::load(b10)), Aff<F1>::load(beta1));
      E1::mul(t, k.v, SW).store(scratch + 3 * J1);
    } break;
    case 4: { FR k = (r * s).from_mont(); E1::mul(Jac<F1>{Aff<F1>::load(delta1).x, Aff<F1>::load(delta1).y, F1::one()}, k.v, SW).store(scratch + 4 * J1); } break;
    case 5: { FR k = s.from_mont(); E2::mul(Jac<F2>{Aff<F2>::load(delta2).x, Aff<F2>::load(delta2).y, F2::one()}, k.v, SW).store(scratch + 5 * J1); } break;
  }
}

__global__ void __launch_bounds__(64) g16_finish(const uint32_t* __restrict__ msm_g1, const uint32_t* __restrict__ msm_g2,
                                                 const uint32_t* __restrict__ singles, const uint32_t* __restrict__ scratch,
                                                 uint32_t* __restrict__ proof) {
  if (threadIdx.x != 0) return;
  typedef EC<GA> E1;
  typedef EC<GB> E2;
  if (blockIdx.x == 0) {  // A = alpha + a0 + r delta + M_a
    Jac<F1> t = E1::add(Jac<F1>::load(scratch), Jac<F1>::load(msm_g1 + 2 * J1));
    t = E1::madd(E1::madd(t, Aff<F1>::load(singles + 3 * A1)), Aff<F1>::load(singles));
    E1::to_affine(t).store(proof);
  } else if (blockIdx.x == 1) {  // B (G2) = beta2 + b2_0 + s delta2 + M_b2
    Jac<F2> t = E2::add(Jac<F2>::load(scratch + 5 * J1), Jac<F2>::load(msm_g2));
    t = E2::madd(E2::madd(t, Aff<F2>::load(singles + 5 * A1 + 2 * A2)), Aff<F2>::load(singles + 5 * A1));
    E2::to_affine(t).store(proof + A1);
  } else {  // C = X3 + X4 + X5 + M_l + M_h
    Jac<F1> t = E1::add(Jac<F1>::load(scratch + 2 * J1), Jac<F1>::load(scratch + 3 * J1));
    t = E1::add(t, Jac<F1>::load(scratch + 4 * J1));
    t = E1::add(t, Jac<F1>::load(msm_g1 + J1));
    t = E1::add(t, Jac<F1>::load(msm_g1));
    E1::to_affine(t).store(proof + A1 + A2);
  }
}

hipError_t assemble(hipStream_t st, const uint32_t* msm_g1, const uint32_t* msm_g2, const uint32_t* singles, const uint32_t* rs_dev,
                    uint32_t* scratch, uint32_t* proof_out) {
  hipLaunchKernelGGL(g16_scalar_muls, dim3(6), dim3(64), 0, st, msm_g1, singles, rs_dev, scratch);
  hipLaunchKernelGGL(g16_finish, dim3(3), dim3(64), 0, st, msm_g1, msm_g2, singles, scratch, proof_out);
  return hipGetLastError();
}

}  // namespace

#define PCD_CAT_(a, b) a##b
#define PCD_CAT(a, b) PCD_CAT_(a, b)
const CurveEntry* PCD_CAT(pcd_curve_entry_, PCD_CURVE_IDX)() {
  static const CurveEntry e = {assemble, (size_t)(5 * J1 + J2) * 4};
  return &e;
}

}  // namespace pcd
