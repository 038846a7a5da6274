// One object per curve (compile with -DPCD_CURVE_IDX=0..3): pairing kernels (pairing.hip.h).
#include "common.h"
#include "pairing.hip.h"

namespace pcd {

#if PCD_CURVE_IDX == 0
typedef PC_MNT4_298 PCT;
#elif PCD_CURVE_IDX == 1
typedef PC_MNT6_298 PCT;
#elif PCD_CURVE_IDX == 2
typedef PC_MNT4_753 PCT;
#elif PCD_CURVE_IDX == 3
typedef PC_MNT6_753 PCT;
#else
#error "PCD_CURVE_IDX must be 0..3"
#endif

namespace {

typedef Pairing<PCT> PE;
typedef typename PE::Fq Fq;
typedef typename PE::E E;
typedef typename PE::Fqk Fqk;
constexpr int A1A = Aff<Fq>::ABI_WORDS, A2A = Aff<E>::ABI_WORDS, GW = Fqk::WORDS, GWA = Fqk::ABI_WORDS;

// one lane per pair: f_i = miller_loop(P_i, Q_i); points arrive in the C-ABI image
__global__ void __launch_bounds__(64) miller_kernel(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                    uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fqk f = PE::miller_loop(Aff<Fq>::from_abi(g1 + (size_t)i * A1A), Aff<E>::from_abi(g2 + (size_t)i * A2A));
  f.store(out + (size_t)i * GW);
}
// one lane per group of `per` consecutive Miller values: their product, then one final exponentiation
__global__ void __launch_bounds__(64) final_exp_kernel(const uint32_t* __restrict__ fs, uint32_t groups, uint32_t per,
                                                       uint32_t* __restrict__ out) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  typename PE::Frob t;
  frob_init<Fq, PE::K, PCT::NR>(t);
  Fqk f = Fqk::one();
  for (uint32_t i = 0; i < per; i++) f = f * Fqk::load(fs + ((size_t)g * per + i) * GW);
  PE::final_exponentiation(f, t).to_abi(out + (size_t)g * GWA);
}

// gt_out[g] = final_exp(prod_{i < per} miller(P_{g per + i}, Q_{g per + i})), g < groups
hipError_t multi_pairing(hipStream_t st, const uint32_t* g1_dev, const uint32_t* g2_dev, uint32_t groups, uint32_t per, uint32_t* scratch,
                         uint32_t* gt_out) {
  const uint32_t n = groups * per;
  if (n) hipLaunchKernelGGL(miller_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g1_dev, g2_dev, n, scratch);
  if (groups) hipLaunchKernelGGL(final_exp_kernel, dim3((groups + 63) / 64), dim3(64), 0, st, scratch, groups, per, gt_out);
  return hipGetLastError();
}

// out[i] = k_i * P_i as a Jacobian point in the C-ABI image; k_i: `kwords` canonical u32 words; P_i affine, C-ABI image ((0, 0) =
// infinity).  One lane per product (the random-linear-combination batch verification scales each proof's A and C by a 128-bit
// challenge: a few hundred group operations of latency next to Miller loops of tens of thousands).
typedef typename PCT::G1 G1C;
__global__ void __launch_bounds__(64) g1_scale_kernel(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ k, uint32_t kwords, uint32_t n,
                                                      uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Aff<Fq> p = Aff<Fq>::from_abi(g1 + (size_t)i * A1A);
  Jac<Fq> r = Jac<Fq>::infinity();
  if (!p.is_inf()) {
    const Jac<Fq> pj = {p.x, p.y, Fq::one()};
    r = EC<G1C>::mul(pj, k + (size_t)i * kwords, (int)kwords);
  }
  if (r.is_inf()) r = Jac<Fq>::infinity();
  r.to_abi(out + (size_t)i * Jac<Fq>::ABI_WORDS);
}
hipError_t g1_scale(hipStream_t st, const uint32_t* g1_dev, const uint32_t* k_dev, uint32_t kwords, uint32_t n, uint32_t* out_dev) {
  if (n) hipLaunchKernelGGL(g1_scale_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g1_dev, k_dev, kwords, n, out_dev);
  return hipGetLastError();
}

}  // namespace

#define PCD_CAT_(a, b) a##b
#define PCD_CAT(a, b) PCD_CAT_(a, b)
const PairingEntry* PCD_CAT(pcd_pairing_entry_, PCD_CURVE_IDX)() {
  static const PairingEntry e = {GWA, GW, multi_pairing};
  return &e;
}
typedef hipError_t (*G1ScaleFn)(hipStream_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*);
G1ScaleFn PCD_CAT(pcd_g1_scale_entry_, PCD_CURVE_IDX)() { return g1_scale; }

}  // namespace pcd
