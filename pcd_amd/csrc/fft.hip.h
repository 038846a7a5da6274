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

#include "ec.hip.h"

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

// First-pass twiddles (round 6).  The twiddle an output of pass i takes is w^(s p k): in every pass but the first the T columns of a tile share
// p, so a workgroup reads R entries of the root table; in the FIRST pass (s = 1) the exponent is idx * k mod n -- a different entry for every
// output, gathered at stride k from the n-entry table: 2^20 gathers of 44 / 108 B that pull whole 128-B lines (rocprofv3, profiles/r06_fft_pass.csv
// before: FETCH x 2 = 257 MB in the first 298-bit pass at 2^20 against 46 MB of vector -- 3.8 TB/s of HBM traffic in 79 us, the one pass of the
// transform that HBM bounds).  The same factors laid out once per (field, n, direction) in the order the pass STORES -- tw0[o], o the output
// index -- are read as one coalesced stream of n elements.
template <class F>
__global__ void __launch_bounds__(256) fft_first_pass_twiddles(const uint32_t* __restrict__ tw, uint32_t* __restrict__ tw0, int logn, int d) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << logn;
  if (o >= n) return;
  const uint32_t idx = o >> d, k = o & ((1u << d) - 1u);
  const uint32_t ex = (uint32_t)(((uint64_t)idx * k) & (n - 1));
  F::load(tw + (size_t)ex * F::WORDS).store(tw0 + (size_t)o * F::WORDS);
}

// Butterfly arithmetic of a pass.  Plain: every addition / subtraction reduces (753-bit fields: R'/p ~ 8 leaves no room).
// Lazy (298-bit fields, R'/p > 2^10): elements travel through the layers of a pass as UNREDUCED signed 28-bit-radix limbs
// (Fp::Lz); a butterfly's sum and difference are limb-wise with no carry chain and no reduction (the difference adds a multiple of
// p that keeps it positive), only the twiddle product reduces, one carry pass per two layers keeps limbs below 2^30.  Value bounds:
// elements enter a pass below 2p and at most double per layer (< 2^(k+1) p after k layers, k <= 7), the difference of two
// elements below B adds K >= B (a multiple of 4p), so a product sees at most 2K * 2p <= 1024 p^2 -- the bound of Fp::lz_mul.
template <class F, bool LAZY = LazyCapable<F>::value>
struct FftArith {
  typedef F E;
  PCD_DEV static E from(const F& v) { return v; }
  PCD_DEV static E ld(const uint32_t* p) { return F::load(p); }
  PCD_DEV static void st(const E& e, uint32_t* p) { e.store(p); }
  PCD_DEV static E add(const E& a, const E& b) { return a + b; }
  struct K {};
  PCD_DEV static K kp(int) { return K(); }
  PCD_DEV static E sub(const E& a, const E& b, const K&) { return a - b; }
  PCD_DEV static E mul_tw(const E& a, const F& w) { return a * w; }
  PCD_DEV static E settle(const E& a) { return a; }
  PCD_DEV static F finish(const E& a, const F& w, bool has_w) { return has_w ? a * w : a; }
};
template <class F>
struct FftArith<F, true> {
  typedef typename F::Lz E;
  typedef typename F::Params P;
  PCD_DEV static E from(const F& v) { return v.lz(); }
  PCD_DEV static E ld(const uint32_t* p) { E r;
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = (int32_t)p[i];
    return r; }
  PCD_DEV static void st(const E& e, uint32_t* p) {
#pragma unroll
    for (int i = 0; i < F::N; i++) p[i] = (uint32_t)e.v[i]; }
  PCD_DEV static E add(const E& a, const E& b) { return F::lz_add(a, b); }
  // (4 << S) p with carry-propagated limbs; a - b + that is non-negative as long as b < (4 << S) p
  typedef E K;
  PCD_DEV static K kp(int S) {
    K r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
      c += (int64_t)P::mod4(i) << S;
      r.v[i] = i < F::N - 1 ? (int32_t)(c & (int64_t)F::MASK) : (int32_t)c;
      c >>= 28;
    }
    return r;
  }
  PCD_DEV static E sub(const E& a, const E& b, const K& k) {
    E r;
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = a.v[i] - b.v[i] + k.v[i];
    return r;
  }
  PCD_DEV static E mul_tw(const E& a, const F& w) { return F::lz_mul(a, w.lz()).lz(); }
  PCD_DEV static E settle(const E& a) { return F::lz_carry(a); }
  PCD_DEV static F finish(const E& a, const F& w, bool) { return F::lz_mul(a, w.lz()); }  // (w = one where there is no twiddle)
};

// One Stockham pass.  x -> y (distinct buffers), tw = powers of the n-th root for this direction.
//   s: stride before this pass (product of earlier radices);  logn: log2 n;  d: log2 R;  logT: log2 T
//   pre  (optional): x_j is multiplied by pre[j] on load (first pass)
//   post (optional): y_o is multiplied by post[o] on store (last pass);  scale (optional flag): by *scale_c
// The d DIF layers along the rows run TWO AT A TIME in registers: a lane owns the four rows base + k 2^l (k < 4) of one column,
// does the layer-(l+1) butterflies (x0, x2), (x1, x3) and the layer-l butterflies on their results, and writes the four values back
// -- half the LDS round trips and barriers of a layer-at-a-time kernel, three twiddle loads per four butterflies, four independent
// chains per lane.  An odd d ends with the twiddle-free layer 0 alone.
template <class F>
__global__ void __launch_bounds__(256) fft_pass_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                       const uint32_t* __restrict__ tw, int logn, int d, int logT, int logs,
                                                       const uint32_t* __restrict__ pre, const uint32_t* __restrict__ post,
                                                       int use_scale, const F scale_c, size_t batch_stride,
                                                       const uint32_t* __restrict__ tw_out = nullptr /* first pass: twiddles in store order */) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  // blockIdx.y: which of several equal transforms laid out `batch_stride` words apart (the rows of a mixed-radix domain go through
  // their radix-2 passes in ONE launch per pass: a 2^14-point row alone is 16 workgroups on a 256-CU chip)
  x += (size_t)blockIdx.y * batch_stride;
  y += (size_t)blockIdx.y * batch_stride;
  typedef FftArith<F> A;
  typedef typename A::E E;
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
    A::st(A::from(v), lds + (size_t)e * EW);
  }
  __syncthreads();
  // d DIF layers along r;  afterwards position r holds b_{bitrev_d(r)}
  int top = d - 1;  // next layer
  int blog = 1;     // elements are below 2^blog p
  auto renormalise = [&]() {  // lazy arithmetic only (single-pass transforms with d > 8): back below 2p before the product bound is reached
    const F one = F::one();
    for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) A::st(A::from(A::finish(A::ld(lds + (size_t)e * EW), one, true)), lds + (size_t)e * EW);
    __syncthreads();
    blog = 1;
  };
  // (plain arithmetic -- the 753-bit fields, whose products are calls: four elements and three twiddles in flight spill -- keeps one
  //  layer per round trip: measured 8 % faster there; the lazy 298-bit arithmetic gains 12-15 % from the two-layer stages)
  while (LazyCapable<F>::value && top >= 1) {
    if (blog > 7) renormalise();
    const int l = top - 1;  // this stage: layers l + 1 and l
    // (4 << S1) p >= 2^blog p,  (4 << S2) p >= 2^(blog + 1) p
    const typename A::K K0 = A::kp(0), K1 = A::kp(blog > 2 ? blog - 2 : 0), K2 = A::kp(blog > 1 ? blog - 1 : 0);
    const uint32_t lmask = (1u << l) - 1u;
    for (uint32_t g = threadIdx.x; g < (tile >> 2); g += blockDim.x) {
      const uint32_t tt = g & (T - 1), jr = g >> logT;  // jr in [0, R / 4)
      const uint32_t jl = jr & lmask, base = ((jr >> l) << (l + 2)) | jl;
      uint32_t* p0 = lds + ((size_t)(base << logT) + tt) * EW;
      const size_t step = ((size_t)1 << (l + logT)) * EW;  // 2^l rows
      E x0 = A::ld(p0), x1 = A::ld(p0 + step), x2 = A::ld(p0 + 2 * step), x3 = A::ld(p0 + 3 * step);
      // layer l + 1 (h = 2^(l+1)): (x0, x2) at position jl, (x1, x3) at position jl + 2^l;  w_{2h}^j = w_n^{j n / (2h)}
      const F w1b = F::load(tw + ((size_t)(jl + (1u << l)) << (logn - l - 2)) * EW);
      E y0 = A::add(x0, x2), y1 = A::add(x1, x3), y2 = A::sub(x0, x2, K1), y3 = A::mul_tw(A::sub(x1, x3, K1), w1b);
      E z0, z1, z2, z3;
      if (l > 0) {
        const F w1a = F::load(tw + ((size_t)jl << (logn - l - 2)) * EW);
        const F w0 = F::load(tw + ((size_t)jl << (logn - l - 1)) * EW);  // layer l: both pairs at position jl
        y2 = A::mul_tw(y2, w1a);
        z1 = A::mul_tw(A::sub(y0, y1, K2), w0);
        z3 = A::mul_tw(A::sub(y2, y3, K0), w0);
      } else {  // jl = 0 for every lane: those twiddles are one
        z1 = A::sub(y0, y1, K2);
        z3 = A::sub(y2, y3, K0);  // (lazy: y2 = x0 - x2 + K1 stays unreduced, below 2 K1; y3 is a product, below 2p)
      }
      z0 = A::settle(A::add(y0, y1));
      z2 = A::settle(A::add(y2, y3));  // (every stored sum is carry-propagated: the next stage adds two more layers on top)
      A::st(z0, p0); A::st(z1, p0 + step); A::st(z2, p0 + 2 * step); A::st(z3, p0 + 3 * step);
    }
    __syncthreads();
    blog += 2;
    top -= 2;
  }
  // the remaining layers one at a time: all of them for the plain arithmetic, the twiddle-free layer 0 of an odd d for the lazy one
  for (; top >= 0; top--) {
    if (LazyCapable<F>::value && blog > 8) renormalise();
    const int layer = top;
    const uint32_t h = 1u << layer;
    const typename A::K Kl = A::kp(blog > 2 ? blog - 2 : 0);
    for (uint32_t bfly = threadIdx.x; bfly < (tile >> 1); bfly += blockDim.x) {
      const uint32_t tt = bfly & (T - 1), jr = bfly >> logT;          // jr in [0, R/2)
      const uint32_t jlow = jr & (h - 1), r0 = ((jr >> layer) << (layer + 1)) | jlow, r1 = r0 + h;
      uint32_t* p0 = lds + ((size_t)(r0 << logT) + tt) * EW;
      uint32_t* p1 = lds + ((size_t)(r1 << logT) + tt) * EW;
      const E a = A::ld(p0), b = A::ld(p1);
      E v = A::sub(a, b, Kl);
      if (jlow) v = A::mul_tw(v, F::load(tw + ((size_t)jlow << (logn - layer - 1)) * EW));  // w_{2h}^{jlow} = w_n^{jlow * n/(2h)}
      A::st(A::settle(A::add(a, b)), p0);
      A::st(v, p1);
    }
    __syncthreads();
    blog += 1;
  }
  // store: y[q + s (R p + k)] = b_k * w_n^{s p k},  idx = i0 + tt = q + s p
  const uint32_t smask = (1u << logs) - 1u;
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    uint32_t k, tt;
    if (logs == 0) { k = e & (R - 1); tt = e >> d; } else { tt = e & (T - 1); k = e >> logT; }
    uint32_t pos = __brev(k) >> (32 - d);
    if (d == 0) pos = 0;
    const E ve = A::ld(lds + ((size_t)(pos << logT) + tt) * EW);
    uint32_t idx = i0 + tt;
    uint32_t q = idx & smask, sp = idx - q;  // s * p
    uint32_t ex = (uint32_t)(((uint64_t)sp * k) & (n - 1));
    uint32_t o = q + ((sp << d) + (k << logs));
    F v = A::finish(ve, F::load(tw_out ? tw_out + (size_t)o * EW : tw + (size_t)ex * EW), ex != 0);
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
