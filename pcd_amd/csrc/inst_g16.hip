// One object per curve (compile with -DPCD_CURVE_IDX=0..3): Groth16 proof assembly (K5 of SURVEY.md
// section 8) -- ark-groth16 `create_proof` after the five MSMs:
//   A = alpha + a_query[0] + r*delta + M_a          B = beta + b_query[0] + s*delta + M_b   (G1 and G2)
//   C = s*A + r*B_1 - r s*delta + M_l + M_h  =  s*(alpha + a_0 + M_a) + r*(beta_1 + b_0 + M_b1) + r s*delta + M_l + M_h
// The two variable-base products s*A and r*B_1 are either folded into MSMs (below) or computed by one lane each on the
// stream of the MSM that produced the point, where they overlap the longer MSMs of the same proof (capi.hip).
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

// Nothing in the assembly is a scalar multiplication (SURVEY.md K5 restated for the GPU): with the key laid out as
//   a'  = a_query  || [delta, O, O, alpha]      b1' = b_g1_query || [O, delta, O, beta_1]
//   b2' = b_g2_query || [O, delta_2, O, beta_2] l'  = l_query || [O, O, delta, O]
// and the scalar tail t1 = [r, s, -rs, 1] after the assignment z (z_0 read as 1):
//   A = MSM(a', z || t1)          B = MSM(b2', z || t1)
//   s*A = MSM(a', s*(z || t1))    r*B_1 = MSM(b1', r*(z || t1))      (the same bases, every scalar scaled)
//   C = s*A + r*B_1 + MSM(l', aux || t1) + MSM(h_query, h)
__global__ void __launch_bounds__(64) g16_prepare_scalars(const uint32_t* __restrict__ rs, uint32_t* __restrict__ t1,
                                                          uint32_t* __restrict__ ts, uint32_t* __restrict__ tr) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  FR r = FR::from_abi(rs), s = FR::from_abi(rs + SWA);
  FR v[4] = {r, s, (r * s).neg(), FR::one()};
  for (int i = 0; i < 4; i++) {
    v[i].to_canonical_words(t1 + i * SWA);
    (v[i] * s).to_canonical_words(ts + i * SWA);
    (v[i] * r).to_canonical_words(tr + i * SWA);
  }
}

__global__ void __launch_bounds__(64) g16_finish(const uint32_t* __restrict__ msm_g1, const uint32_t* __restrict__ msm_g2,
                                                 uint32_t* __restrict__ proof_abi) {
  if (threadIdx.x != 0) return;
  typedef EC<GA> E1;
  typedef EC<GB> E2;
  if (blockIdx.x == 0) {
    E1::to_affine(Jac<F1>::load(msm_g1 + 2 * J1)).to_abi(proof_abi);
  } else if (blockIdx.x == 1) {
    E2::to_affine(Jac<F2>::load(msm_g2)).to_abi(proof_abi + A1A);
  } else {
    Jac<F1> t = E1::add(Jac<F1>::load(msm_g1 + 3 * J1), Jac<F1>::load(msm_g1 + 4 * J1));
    t = E1::add(t, Jac<F1>::load(msm_g1 + J1));
    t = E1::add(t, Jac<F1>::load(msm_g1));
    E1::to_affine(t).to_abi(proof_abi + A1A + A2A);
  }
}

// out = k * in (k: canonical words).  Round 4's form was ONE lane (4-bit window: ~300 / 750 doublings and ~75 / 190 additions of 9 .. 16
// dependent products each: 4.6 ms over the 298-bit fields, ~60 ms over the 753-bit ones -- the longest kernel of a proof, harmless only
// while it hid under three other accumulations).  Now one workgroup of four waves:
//   1. lanes 0 .. 3 run the doubling chain, FOUR lanes per doubling in XYZZ coordinates (dbl-2008-s-1: its ten products are three levels deep --
//      V = (2Y)^2, X^2, ZZ^2 | W = 2Y V, S = X V, M^2, ZZ' = V ZZ | M (S - X'), W Y, ZZZ' = W ZZZ -- so three product slots a doubling where
//      the two-lane Jacobian form (EC2::dbl2) has five: the chain is the kernel's critical path, bits of the scalar field many) and leave
//      T_j = 16^j P behind for every 4-bit window j (four coordinates; the readers convert);
//   2. every lane PAIR takes windows j, j + 128, ..: d_j T_j by double-and-add over the digit's four bits (the pairs of a wave run in
//      lockstep, so a pair pays the four doublings and four additions whatever its digit), summed per pair;
//   3. a tree over the 128 pairs through `scratch`.
// A witness-like proof waits for exactly this kernel behind its sparse A / B_1 MSMs (profiles/r05_witness_like_critical_path.txt).
// scratch: SCALE_SLOTS Jacobian slots (window table of four-coordinate records, then one partial per pair).
constexpr int SCALE_W = 4, SCALE_NW = (GA::FR::BITS + SCALE_W - 1) / SCALE_W, SCALE_PAIRS = 128;
constexpr int SCALE_TABLE_SLOTS = (SCALE_NW * 4 + 2) / 3, SCALE_SLOTS = SCALE_TABLE_SLOTS + SCALE_PAIRS;
static_assert(SCALE_SLOTS <= 400, "G16Run::prepare reserves 400 Jacobian slots per product");
struct ScaleQuad {  // four lanes per doubling
  typedef F1 F;
  struct P4 { F X, Y, ZZ, ZZZ; };
  template <int K> PCD_DEV static F bc(const F& a) { F r;   // lane K of every quad to all four of its lanes
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], K * 0x55, 0xF, 0xF, false);
    return r; }
  PCD_DEV static F sel4(uint32_t l, const F& a0, const F& a1, const F& a2, const F& a3) { F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) { const uint32_t lo = (l & 1u) ? a1.v[i] : a0.v[i], hi = (l & 1u) ? a3.v[i] : a2.v[i]; r.v[i] = (l & 2u) ? hi : lo; }
    return r; }
  PCD_DEV static P4 dbl4(const P4& p) {  // (a finite point of odd prime order: no doubling of this chain meets the identity)
    const uint32_t l = threadIdx.x & 3u;
    const F U = p.Y.dbl();
    const F a1 = sel4(l, U, p.X, p.ZZ, U);
    const F s1 = a1 * a1;                                   // lane 0: V = U^2      1: X^2      2: ZZ^2     (3: idle)
    const F V = bc<0>(s1), XX = bc<1>(s1), Z4 = bc<2>(s1);
    const F M = XX.dbl() + XX + GA::mul_by_a(Z4);
    const F a2 = sel4(l, U, p.X, M, V), b2 = sel4(l, V, V, M, p.ZZ);
    const F s2 = a2 * b2;                                   // lane 0: W = U V      1: S = X V  2: M^2      3: ZZ' = V ZZ
    const F W = bc<0>(s2), S = bc<1>(s2), MM = bc<2>(s2);
    P4 r;
    r.ZZ = bc<3>(s2);
    r.X = MM - S.dbl();
    const F smx = S - r.X;
    const F a3 = sel4(l, M, W, W, M), b3 = sel4(l, smx, p.Y, p.ZZZ, smx);
    const F s3 = a3 * b3;                                   // lane 0: M (S - X')   1: W Y      2: ZZZ' = W ZZZ     (3: idle)
    r.Y = bc<0>(s3) - bc<1>(s3);
    r.ZZZ = bc<2>(s3);
    return r;
  }
};
__global__ void __launch_bounds__(2 * SCALE_PAIRS) g16_scale_point(const uint32_t* __restrict__ in, const uint32_t* __restrict__ k,
                                                                  uint32_t* __restrict__ scratch, uint32_t* __restrict__ out) {
  if (blockIdx.x != 0) return;
  typedef EC2<GA> E2;
  typedef Jac<F1> J;
  constexpr int FW = F1::WORDS;
  const uint32_t pair = threadIdx.x >> 1;
  const bool writer = (threadIdx.x & 1u) == 0;
  uint32_t* table = scratch;
  uint32_t* partial = scratch + (size_t)SCALE_TABLE_SLOTS * J1;
  const J p_in = J::load(in);
  if (p_in.is_inf()) {  // (uniform: every lane reads the same point)
    if (threadIdx.x == 0) J::infinity().store(out);
    return;
  }
  if (threadIdx.x < 4) {
    const F1 zz = p_in.Z.sqr();
    ScaleQuad::P4 t = {p_in.X, p_in.Y, zz, zz * p_in.Z};
    for (int j = 0; j < SCALE_NW; j++) {
      if (threadIdx.x == 0) { t.X.store(table + (size_t)j * 4 * FW); t.Y.store(table + ((size_t)j * 4 + 1) * FW); t.ZZ.store(table + ((size_t)j * 4 + 2) * FW); t.ZZZ.store(table + ((size_t)j * 4 + 3) * FW); }
      if (j + 1 < SCALE_NW)
        for (int d = 0; d < SCALE_W; d++) t = ScaleQuad::dbl4(t);
    }
  }
  __syncthreads();
  J acc = J::infinity();
  for (uint32_t j = pair; j < (uint32_t)SCALE_NW; j += SCALE_PAIRS) {
    const uint32_t bit0 = j * SCALE_W;
    const uint32_t dgt = bit0 / 32 < (uint32_t)SWA ? (k[bit0 / 32] >> (bit0 % 32)) & 15u : 0u;  // (4 divides 32: a digit never straddles words)
    // the Jacobian point (X ZZ : Y ZZZ : ZZ) of the record: one product slot of the pair
    const uint32_t* rec = table + (size_t)j * 4 * FW;
    const F1 zz = F1::load(rec + 2 * FW);
    const F1 pr = E2::slot(F1::load(rec), zz, F1::load(rec + FW), F1::load(rec + 3 * FW));
    const F1 prx = E2::xch(pr);
    const J t = {E2::sel(E2::odd(), prx, pr), E2::sel(E2::odd(), pr, prx), zz};
    J r = J::infinity();
    for (int b = SCALE_W - 1; b >= 0; b--) {
      r = E2::dbl2(r);
      if ((dgt >> b) & 1u) r = E2::add2(r, t);
    }
    acc = E2::add2(acc, r);
  }
  if (writer) acc.store(partial + (size_t)pair * J1);
  __syncthreads();
  for (uint32_t s = SCALE_PAIRS / 2; s > 0; s >>= 1) {
    if (pair < s) {
      acc = E2::add2(acc, J::load(partial + (size_t)(pair + s) * J1));
      if (writer) acc.store(partial + (size_t)pair * J1);
    }
    __syncthreads();
  }
  if (pair == 0 && writer) acc.store(out);
}

hipError_t scale_g1(hipStream_t st, const uint32_t* in, const uint32_t* k, uint32_t* scratch, uint32_t* out) {
  hipLaunchKernelGGL(g16_scale_point, dim3(1), dim3(2 * SCALE_PAIRS), 0, st, in, k, scratch, out);
  return hipGetLastError();
}
hipError_t prepare_scalars(hipStream_t st, const uint32_t* rs_dev, uint32_t* t1, uint32_t* ts, uint32_t* tr) {
  hipLaunchKernelGGL(g16_prepare_scalars, dim3(1), dim3(64), 0, st, rs_dev, t1, ts, tr);
  return hipGetLastError();
}
hipError_t assemble(hipStream_t st, const uint32_t* msm_g1, const uint32_t* msm_g2, uint32_t* proof_out) {
  hipLaunchKernelGGL(g16_finish, dim3(3), dim3(64), 0, st, msm_g1, msm_g2, proof_out);
  return hipGetLastError();
}

}  // namespace

#define PCD_CAT_(a, b) a##b
#define PCD_CAT(a, b) PCD_CAT_(a, b)
const CurveEntry* PCD_CAT(pcd_curve_entry_, PCD_CURVE_IDX)() {
  static const CurveEntry e = {prepare_scalars, (size_t)(2 * A1A + A2A) * 4, assemble, scale_g1};
  return &e;
}

}  // namespace pcd
