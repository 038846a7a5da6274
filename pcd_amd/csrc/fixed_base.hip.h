// Fixed-base batch multiplication  out[i] = k_i * B  with affine results.
//
// Device counterpart of ark-ec `FixedBaseMSM::{get_window_table, multi_scalar_mul}` followed by
// `ProjectiveCurve::batch_normalization_into_affine`, the primitive ark-groth16 `generate_parameters` spends its time
// in (every query of the proving key is such a batch) -- reached from the reference through
// `circuit_specific_setup` (/root/reference src/ec_cycle_pcd/mod.rs:69,78); the KZG powers of `universal_setup`
// (mod.rs:346-354) are the same primitive.  SURVEY.md section 8(f) rank 2.
//
// Layout: window table T[j][d] = d * 2^(w j) * B as affine points in the device image (nwin * 2^w entries, entry d = 0
// unused); one lane per scalar adds its nwin table entries with mixed additions; a second kernel normalises FB_BATCH
// Jacobian results per lane with one inversion (Montgomery's trick) and writes the C-ABI affine image + flags.
#pragma once
#include "ec.hip.h"

namespace pcd {

constexpr int FB_WINDOW = 8;

// B_j = 2^(w j) B, one lane (a chain of doublings).  One workgroup per base: base b is the b-th C-ABI point behind base_abi, its chain
// goes to bj + b * slot_stride -- the chains of several bases (the public-input bases of a verifying key) run side by side instead
// of one launch after the other.
template <class G>
__global__ void __launch_bounds__(64) fb_powers_kernel(const uint32_t* __restrict__ base_abi, uint32_t* __restrict__ bj, int nwin, int w, size_t slot_stride) {
  if (threadIdx.x != 0) return;
  typedef typename G::F F;
  base_abi += (size_t)blockIdx.x * Aff<F>::ABI_WORDS;
  bj += (size_t)blockIdx.x * slot_stride;
  Aff<F> b = Aff<F>::from_abi(base_abi);
  Jac<F> p = b.is_inf() ? Jac<F>::infinity() : Jac<F>{b.x, b.y, F::one()};
  for (int j = 0; j < nwin; j++) {
    p.store(bj + (size_t)j * Jac<F>::WORDS);
    for (int k = 0; k < w; k++) p = EC<G>::dbl(p);
  }
}

// T[j][d] = d * B_j (w-bit double-and-add), affine; blockIdx.y = the base (chains and tables slot_stride words apart)
template <class G>
__global__ void __launch_bounds__(64) fb_table_kernel(const uint32_t* __restrict__ bj, uint32_t* __restrict__ table, int nwin, int w, size_t slot_stride) {
  typedef typename G::F F;
  bj += (size_t)blockIdx.y * slot_stride;
  table += (size_t)blockIdx.y * slot_stride;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = t >> w, d = t & ((1u << w) - 1);
  if (j >= (uint32_t)nwin || d == 0) return;
  const Jac<F> p = Jac<F>::load(bj + (size_t)j * Jac<F>::WORDS);
  Jac<F> r = Jac<F>::infinity();
  for (int k = w - 1; k >= 0; k--) {
    r = EC<G>::dbl(r);
    if ((d >> k) & 1) r = EC<G>::add(r, p);
  }
  EC<G>::to_affine(r).store(table + (size_t)t * Aff<F>::WORDS);
}

// acc_i = sum_j T[j][digit_j(k_i)]  (Jacobian, device image)
template <class G>
__global__ void __launch_bounds__(64) fb_mul_kernel(const uint32_t* __restrict__ table, const uint32_t* __restrict__ scalars, uint32_t n,
                                                    int nwin, int w, uint32_t* __restrict__ out_jac) {
  typedef typename G::F F;
  constexpr int SW = G::FR::N32;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* k = scalars + (size_t)i * SW;
  Jac<F> acc = Jac<F>::infinity();
  for (int j = 0; j < nwin; j++) {
    const int bit = j * w, word = bit >> 5, sh = bit & 31;
    uint32_t d = k[word] >> sh;
    if (sh + w > 32 && word + 1 < SW) d |= k[word + 1] << (32 - sh);
    d &= (1u << w) - 1;
    if (d) acc = EC<G>::madd(acc, Aff<F>::load(table + (((size_t)j << w) + d) * Aff<F>::WORDS));
  }
  acc.store(out_jac + (size_t)i * Jac<F>::WORDS);
}

// Jacobian (device image) -> affine C-ABI image + infinity flags; FB_BATCH points per lane share one inversion
constexpr int FB_BATCH = 8;
template <class G>
__global__ void __launch_bounds__(64) fb_normalize_kernel(const uint32_t* __restrict__ jac, uint32_t n, uint32_t* __restrict__ out_abi,
                                                          uint8_t* __restrict__ out_inf) {
  typedef typename G::F F;
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = (size_t)lane * FB_BATCH;
  if (lo >= n) return;
  const int cnt = (int)((n - lo) < (size_t)FB_BATCH ? (n - lo) : (size_t)FB_BATCH);
  F pre[FB_BATCH];  // pre[k] = product of the non-zero Z of points 0..k-1
  F run = F::one();
  for (int k = 0; k < cnt; k++) {
    pre[k] = run;
    const F z = F::load(jac + (lo + k) * Jac<F>::WORDS + 2 * F::WORDS);
    if (!z.is_zero()) run = run * z;
  }
  F inv = run.inv();
  for (int k = cnt - 1; k >= 0; k--) {
    const Jac<F> p = Jac<F>::load(jac + (lo + k) * Jac<F>::WORDS);
    uint32_t* o = out_abi + (lo + k) * Aff<F>::ABI_WORDS;
    if (p.Z.is_zero()) {
      Aff<F>{F::zero(), F::zero()}.to_abi(o);
      out_inf[lo + k] = 1;
      continue;
    }
    const F zi = inv * pre[k];
    inv = inv * p.Z;
    const F zi2 = zi.sqr();
    Aff<F>{p.X * zi2, p.Y * zi2 * zi}.to_abi(o);
    out_inf[lo + k] = 0;
  }
}

// ---- prepared public inputs of a verification: acc_i = abc_0 + sum_{j >= 1} x_ij abc_j with the abc_j FIXED (a verifying key:
// `prepare_verifying_key` / the input accumulation of ark-groth16 `prepare_inputs`, under /root/reference src/ec_cycle_pcd/mod.rs:239).
// As a variable-base MSM of a handful of points this was a chain of ~300 (750) dependent doublings in one lane -- 5.8 ms (37 ms) per
// proof, more than the pairings of the verification once those run one wave each.  With a window table per abc_j (built once, at
// process_vk) it is (ni - 1) nwin mixed additions spread over the 64 lanes of one workgroup per proof, a tree over the lanes and
// one inversion: tables[j] = the fb_table_kernel table of abc_j, j = 1 .. ni - 1.
template <class G>
__global__ void __launch_bounds__(64) fb_inputs_kernel(const uint32_t* __restrict__ tables, size_t tab_words, const uint32_t* __restrict__ abc0_abi, uint32_t ni,
                                                       const uint32_t* __restrict__ scalars /* k x (ni - 1) canonical */, int nwin, int w,
                                                       uint32_t* __restrict__ scratch /* k x 64 Jacobians */, uint32_t* __restrict__ out_abi,
                                                       uint8_t* __restrict__ out_inf, uint32_t* __restrict__ out_z_abi /* nullable */,
                                                       uint32_t out_stride /* points between two proofs' results in out_abi / out_z_abi */) {
  typedef typename G::F F;
  constexpr int SW = G::FR::N32, JW = Jac<F>::WORDS;
  const uint32_t proof = blockIdx.x, lane = threadIdx.x;
  Jac<F> acc = Jac<F>::infinity();
  if (lane == 0) { const Aff<F> a0 = Aff<F>::from_abi(abc0_abi); if (!a0.is_inf()) acc = {a0.x, a0.y, F::one()}; }
  const uint32_t items = (ni - 1) * (uint32_t)nwin;
  for (uint32_t it = lane; it < items; it += 64) {
    const uint32_t j = it / (uint32_t)nwin, win = it % (uint32_t)nwin;
    const uint32_t* k = scalars + ((size_t)proof * (ni - 1) + j) * SW;
    const int bit = (int)win * w, word = bit >> 5, sh = bit & 31;
    uint32_t d = k[word] >> sh;
    if (sh + w > 32 && word + 1 < SW) d |= k[word + 1] << (32 - sh);
    d &= (1u << w) - 1;
    if (d) acc = EC<G>::madd(acc, Aff<F>::load(tables + (size_t)j * tab_words + (((size_t)win << w) + d) * Aff<F>::WORDS));
  }
  uint32_t* my = scratch + (size_t)proof * 64 * JW;
  acc.store(my + (size_t)lane * JW);
  __syncthreads();
  for (uint32_t s = 32; s > 0; s >>= 1) {
    if (lane < s) {
      acc = EC<G>::add(acc, Jac<F>::load(my + (size_t)(lane + s) * JW));
      acc.store(my + (size_t)lane * JW);
    }
    __syncthreads();
  }
  if (lane == 0) {
    out_inf[proof] = acc.is_inf() ? 1 : 0;
    if (out_z_abi) {  // Jacobian result (X, Y | Z): the pairing VM takes it as it is -- no inversion (pairing_vm.hip.h)
      if (acc.is_inf()) acc = Jac<F>::infinity();
      const Aff<F> xy = {acc.X, acc.Y};
      xy.to_abi(out_abi + (size_t)proof * out_stride * Aff<F>::ABI_WORDS);
      acc.Z.to_abi(out_z_abi + (size_t)proof * out_stride * F::ABI_WORDS);
    } else {
      EC<G>::to_affine(acc).to_abi(out_abi + (size_t)proof * out_stride * Aff<F>::ABI_WORDS);
    }
  }
}
// tables of the bases 1 .. ni - 1 (C-ABI affine image, consecutive), `tab_words` apart: each slot = the table, then the nwin Jacobians
// of its doubling chain
template <class G>
hipError_t fb_tables_build(hipStream_t st, const uint32_t* bases_abi, uint32_t ni, size_t tab_words, uint32_t* tables) {
  constexpr int w = FB_WINDOW, nwin = (G::FR::BITS + w - 1) / w;
  const size_t pure = ((size_t)nwin << w) * Aff<typename G::F>::WORDS;
  if (ni > 1) {  // every base's doubling chain at once, then every table (was: ni - 1 pairs of launches, the chains one after the other)
    hipLaunchKernelGGL((fb_powers_kernel<G>), dim3(ni - 1), dim3(64), 0, st, bases_abi + Aff<typename G::F>::ABI_WORDS, tables + pure, nwin, w, tab_words);
    hipLaunchKernelGGL((fb_table_kernel<G>), dim3(((nwin << w) + 63) / 64, ni - 1), dim3(64), 0, st, tables + pure, tables, nwin, w, tab_words);
  }
  return hipGetLastError();
}
template <class G>
hipError_t fb_inputs_run(hipStream_t st, const uint32_t* tables, size_t tab_words, const uint32_t* abc0_abi, uint32_t ni, const uint32_t* scalars,
                         uint32_t k, uint32_t* scratch, uint32_t* out_abi, uint8_t* out_inf, uint32_t* out_z_abi, uint32_t out_stride) {
  constexpr int w = FB_WINDOW, nwin = (G::FR::BITS + w - 1) / w;
  if (k) hipLaunchKernelGGL((fb_inputs_kernel<G>), dim3(k), dim3(64), 0, st, tables, tab_words, abc0_abi, ni, scalars, nwin, w, scratch, out_abi, out_inf, out_z_abi, out_stride);
  return hipGetLastError();
}

// table: nwin << w affine points; bj: nwin Jacobians; jac_tmp: n Jacobians (all device image)
template <class G>
hipError_t fixed_base_run(hipStream_t st, const uint32_t* base_abi, const uint32_t* scalars, uint32_t n, uint32_t* bj, uint32_t* table,
                          uint32_t* jac_tmp, uint32_t* out_abi, uint8_t* out_inf) {
  constexpr int w = FB_WINDOW, nwin = (G::FR::BITS + w - 1) / w;
  hipLaunchKernelGGL((fb_powers_kernel<G>), dim3(1), dim3(64), 0, st, base_abi, bj, nwin, w, (size_t)0);
  hipLaunchKernelGGL((fb_table_kernel<G>), dim3(((nwin << w) + 63) / 64), dim3(64), 0, st, bj, table, nwin, w, (size_t)0);
  if (n) {
    hipLaunchKernelGGL((fb_mul_kernel<G>), dim3((n + 63) / 64), dim3(64), 0, st, table, scalars, n, nwin, w, jac_tmp);
    const uint32_t lanes = (n + FB_BATCH - 1) / FB_BATCH;
    hipLaunchKernelGGL((fb_normalize_kernel<G>), dim3((lanes + 63) / 64), dim3(64), 0, st, jac_tmp, n, out_abi, out_inf);
  }
  return hipGetLastError();
}

}  // namespace pcd
