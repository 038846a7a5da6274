// Radix-2 FFT / iFFT over the MNT scalar fields on gfx950 (K2 of SURVEY.md section 8).
//
// Replaces ark-poly `Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place` as used by
// Groth16's `R1CSToQAP::witness_map` under SNARK::prove (/root/reference
// src/ec_cycle_pcd/mod.rs:171,179).  Natural order in, natural order out, same domain definition:
// omega = TWO_ADIC_ROOT ^ (2^(s - log n)), coset shift by the field's multiplicative generator.
//
// Structure: Stockham autosort, ceil(log n / 7) HBM passes.  A pass of radix R = 2^d is done by
// workgroups that own a tile of R x T elements (T consecutive columns, R rows at stride n/R): the
// tile is read with T-element contiguous segments, the d butterfly layers run in LDS, and the
// twiddled outputs leave as T-element (or, in the first pass, R*T-element) contiguous segments.
// Coset pre-scaling is fused into the first pass' loads and 1/n / coset post-scaling into the
// last pass' stores, so a transform is exactly `passes` reads and writes of the vector.
#pragma once
#include <vector>

#include "fp.hip.h"

namespace pcd {

struct FftPass { int d; int logT; };

inline std::vector<FftPass> fft_plan(int L) {
  std::vector<FftPass> p;
  if (L <= 10) { p.push_back({L, 0}); return p; }
  int P = (L + 6) / 7;
  for (int i = 0; i < P; i++) {
    int d = L / P + (i < L % P ? 1 : 0);
    p.push_back({d, 10 - d});
  }
  return p;
}

// tw[i] = w^i (i < n), Montgomery form.  One thread per 256-entry run.
template <class F>
__global__ void fft_fill_powers(uint32_t* __restrict__ tw, uint32_t n, const F base, const F first) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t lo = t * 256;
  if (lo >= n) return;
  F cur = first * base.pow_u64(lo);
  for (uint32_t i = lo; i < n && i < lo + 256; i++) { cur.store(tw + (size_t)i * F::WORDS); cur = cur * base; }
}

// One Stockham pass.  x -> y (distinct buffers), tw = powers of the n-th root for this direction.
//   s: stride before this pass (product of earlier radices);  logn: log2 n;  d: log2 R;  logT: log2 T
//   pre  (optional): x_j is multiplied by pre[j] on load (first pass)
//   post (optional): y_o is multiplied by post[o] on store (last pass);  scale (optional flag): by *scale_c
template <class F>
__global__ void __launch_bounds__(256) fft_pass_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                       const uint32_t* __restrict__ tw, int logn, int d, int logT, int logs,
                                                       const uint32_t* __restrict__ pre, const uint32_t* __restrict__ post,
                                                       int use_scale, const F scale_c) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  constexpr int EW = F::WORDS;
  const uint32_t R = 1u << d, T = 1u << logT;
  const uint32_t n = 1u << logn;
  const uint32_t nR = n >> d;  // n / R
  const uint32_t i0 = blockIdx.x << logT;
  const uint32_t tile = R << logT;
  // load: tile[r][tt] = x[i0 + tt + r * n/R]
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    uint32_t r = e >> logT, tt = e & (T - 1);
    uint32_t j = i0 + tt + r * nR;
    F v = F::load(x + (size_t)j * EW);
    if (pre) v = v * F::load(pre + (size_t)j * EW);
    v.store(lds + (size_t)e * EW);
  }
  __syncthreads();
  // d DIF layers along r;  afterwards position r holds b_{bitrev_d(r)}
  for (int layer = d - 1; layer >= 0; layer--) {
    const uint32_t h = 1u << layer;
    for (uint32_t bfly = threadIdx.x; bfly < (tile >> 1); bfly += blockDim.x) {
      uint32_t tt = bfly & (T - 1), jr = bfly >> logT;          // jr in [0, R/2)
      uint32_t jlow = jr & (h - 1), r0 = ((jr >> layer) << (layer + 1)) | jlow, r1 = r0 + h;
      uint32_t* p0 = lds + ((size_t)(r0 << logT) + tt) * EW;
      uint32_t* p1 = lds + ((size_t)(r1 << logT) + tt) * EW;
      F a = F::load(p0), b = F::load(p1);
      F u = a + b, v = a - b;
      if (jlow) v = v * F::load(tw + ((size_t)jlow << (logn - layer - 1)) * EW);  // w_{2h}^{jlow} = w_n^{jlow * n/(2h)}
      u.store(p0);
      v.store(p1);
    }
    __syncthreads();
  }
  // store: y[q + s (R p + k)] = b_k * w_n^{s p k},  idx = i0 + tt = q + s p
  const uint32_t smask = (1u << logs) - 1u;
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    uint32_t k, tt;
    if (logs == 0) { k = e & (R - 1); tt = e >> d; } else { tt = e & (T - 1); k = e >> logT; }
    uint32_t pos = __brev(k) >> (32 - d);
    if (d == 0) pos = 0;
    F v = F::load(lds + ((size_t)(pos << logT) + tt) * EW);
    uint32_t idx = i0 + tt;
    uint32_t q = idx & smask, sp = idx - q;  // s * p
    uint32_t ex = (uint32_t)(((uint64_t)sp * k) & (n - 1));
    if (ex) v = v * F::load(tw + (size_t)ex * EW);
    uint32_t o = q + ((sp << d) + (k << logs));
    if (post) v = v * F::load(post + (size_t)o * EW);
    if (use_scale) v = v * scale_c;
    v.store(y + (size_t)o * EW);
  }
}

// out[i] = (a[i] * b[i] - c[i]) * k     (witness-map pointwise step on the coset)
template <class F>
__global__ void __launch_bounds__(256) fft_mul_sub_scale(uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                         const uint32_t* __restrict__ c, uint32_t n, const F k) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F r = (F::load(a + (size_t)i * F::WORDS) * F::load(b + (size_t)i * F::WORDS) - F::load(c + (size_t)i * F::WORDS)) * k;
  r.store(a + (size_t)i * F::WORDS);
}

// ---- mixed-radix domains n = m * 2^a, m = q or q^2 (ark-poly MixedRadixEvaluationDomain, K2m of SURVEY.md 8a) ----
// X[k1 + m k2] = sum_{j2} w_n^{j2 k1} [ sum_{j1} x[N2 j1 + j2] w_m^{j1 k1} ] w_{N2}^{j2 k2},  N2 = 2^a:
//   step 1 (this kernel): m-point DFTs down the columns + twiddle, y[k1][j2] row-major;
//   step 2: the m rows go through the radix-2 passes above (root w_n^m);  step 3: interleaving store.
// One lane per column j2; the m x m inner products re-read the column from L2.
template <class F>
__global__ void __launch_bounds__(64) fft_mixed_columns_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                               const uint32_t* __restrict__ tw /* w_n^j, j < N2 */, uint32_t N2, uint32_t m,
                                                               const F wm, const uint32_t* __restrict__ pre) {
  constexpr int EW = F::WORDS;
  uint32_t j2 = blockIdx.x * blockDim.x + threadIdx.x;
  if (j2 >= N2) return;
  F t = F::load(tw + (size_t)j2 * EW);
  F twk = F::one(), wk = F::one();
  for (uint32_t k1 = 0; k1 < m; k1++) {
    F acc = F::zero(), pw = F::one();
    for (uint32_t j1 = 0; j1 < m; j1++) {
      size_t j = (size_t)N2 * j1 + j2;
      F v = F::load(x + j * EW);
      if (pre) v = v * F::load(pre + j * EW);
      acc = acc + (j1 ? v * pw : v);
      pw = pw * wk;
    }
    if (k1) acc = acc * twk;
    acc.store(y + ((size_t)k1 * N2 + j2) * EW);
    twk = twk * t;
    wk = wk * wm;
  }
}
// step 3: out[k1 + m k2] = z[k1][k2] (* post[o]) (* scale)
template <class F>
__global__ void __launch_bounds__(256) fft_mixed_interleave_kernel(const uint32_t* __restrict__ z, uint32_t* __restrict__ out, uint32_t N2,
                                                                   uint32_t m, const uint32_t* __restrict__ post, int use_scale, const F scale_c) {
  constexpr int EW = F::WORDS;
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= N2 * m) return;
  uint32_t k1 = o % m, k2 = o / m;
  F v = F::load(z + ((size_t)k1 * N2 + k2) * EW);
  if (post) v = v * F::load(post + (size_t)o * EW);
  if (use_scale) v = v * scale_c;
  v.store(out + (size_t)o * EW);
}

// C-ABI <-> device image of field-element vectors.  MODE 0: ABI Montgomery -> internal, 1: internal -> ABI
// Montgomery, 2: internal -> canonical words (`into_repr()`, the scalars handed to the MSM), 3: ABI Montgomery ->
// canonical words.
template <class F, int MODE>
__global__ void __launch_bounds__(256) fp_convert_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (MODE == 0) F::from_abi(in + (size_t)i * F::ABI_WORDS).store(out + (size_t)i * F::WORDS);
  else if (MODE == 1) F::load(in + (size_t)i * F::WORDS).to_abi(out + (size_t)i * F::ABI_WORDS);
  else if (MODE == 2) F::load(in + (size_t)i * F::WORDS).to_canonical_words(out + (size_t)i * F::ABI_WORDS);
  else F::from_abi(in + (size_t)i * F::ABI_WORDS).to_canonical_words(out + (size_t)i * F::ABI_WORDS);
}

}  // namespace pcd
