// One object per scalar field (compile with -DPCD_FIELD_IDX=0..3): FFT driver, witness-map helpers.
#include "common.h"
#include <string.h>

#include "fft.hip.h"

namespace pcd {

#if PCD_FIELD_IDX == 0
typedef Fp<F298A> FT;
#elif PCD_FIELD_IDX == 1
typedef Fp<F298B> FT;
#elif PCD_FIELD_IDX == 2
typedef Fp<F753A> FT;
#elif PCD_FIELD_IDX == 3
typedef Fp<F753B> FT;
#else
#error "PCD_FIELD_IDX must be 0..3"
#endif

// The transform passes compute with the product INLINED for every field (PCD_FFT_INLINE_753=0: the 753-bit passes through the
// non-inlined call as in round 3): a pass has five product sites, not the dozens of a point kernel, so the 1 458-multiply-add body of the
// 27-limb product costs neither compile time nor the instruction cache here, and the scratch round trips of the call (54 words out, 27
// back, per product, at one wave per SIMD) go away.  Same memory image; only the kernel's arithmetic type differs.
#ifndef PCD_FFT_INLINE_753
#define PCD_FFT_INLINE_753 1
#endif
#if PCD_FFT_INLINE_753
typedef Fp<FT::Params, true> FTP;
#else
typedef FT FTP;
#endif
static_assert(sizeof(FTP) == sizeof(FT), "same image");

namespace {

constexpr int EW = FT::WORDS;

// Host-side copies of a few field constants are produced on the DEVICE (no host bigint code in the
// product): this kernel writes w, w^-1, g, g^-1, 1/n, 1/Z(g) for a domain of 2^log_n.
__global__ void domain_consts_kernel(int log_n, uint32_t* out /* 6 elements */) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  FT w = FT::two_adic_root();
  for (int i = log_n; i < FT::Params::TWO_ADICITY; i++) w = w.sqr();
  FT g = FT::generator();
  FT ninv = FT::from_u64(1ull << log_n).inv();
  FT gn = g;
  for (int i = 0; i < log_n; i++) gn = gn.sqr();
  FT zinv = (gn - FT::one()).inv();
  w.store(out);
  w.inv().store(out + EW);
  g.store(out + 2 * EW);
  g.inv().store(out + 3 * EW);
  ninv.store(out + 4 * EW);
  zinv.store(out + 5 * EW);
}

struct DomainConsts { FT w, winv, g, ginv, ninv, zinv; };
inline bool rem_check(uint32_t n, uint32_t m) { return m == 0 || n % m != 0 || ((n / m) & (n / m - 1)) != 0; }

hipError_t get_consts(hipStream_t st, int log_n, DomainConsts* c) {
  uint32_t* d = nullptr;
  PCD_HIP_TRY(hipMalloc(&d, 6 * EW * 4));
  hipLaunchKernelGGL(domain_consts_kernel, dim3(1), dim3(64), 0, st, log_n, d);
  hipError_t e = hipMemcpyAsync(c, d, 6 * EW * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  return e;
}

hipError_t make_tables(hipStream_t st, int log_n, FftTables* t) {
  DomainConsts c;
  PCD_HIP_TRY(get_consts(st, log_n, &c));
  static_assert(sizeof(DomainConsts) <= sizeof(t->consts), "consts buffer too small");
  memcpy(t->consts, &c, sizeof c);
  const uint32_t n = 1u << log_n;
  const size_t bytes = (size_t)n * EW * 4;
  PCD_HIP_TRY(hipMalloc(&t->tw_fwd, bytes));
  PCD_HIP_TRY(hipMalloc(&t->tw_inv, bytes));
  PCD_HIP_TRY(hipMalloc(&t->coset, bytes));
  PCD_HIP_TRY(hipMalloc(&t->coset_inv_scaled, bytes));
  dim3 gd(((n + 255) / 256 + 63) / 64), bd(64);
  hipLaunchKernelGGL(fft_fill_powers<FT>, gd, bd, 0, st, t->tw_fwd, n, c.w, FT::one());
  hipLaunchKernelGGL(fft_fill_powers<FT>, gd, bd, 0, st, t->tw_inv, n, c.winv, FT::one());
  hipLaunchKernelGGL(fft_fill_powers<FT>, gd, bd, 0, st, t->coset, n, c.g, FT::one());
  hipLaunchKernelGGL(fft_fill_powers<FT>, gd, bd, 0, st, t->coset_inv_scaled, n, c.ginv, c.ninv);  // g^-j / n
  const std::vector<FftPass> plan = fft_plan(log_n);
  if (plan.size() > 1) {  // the first pass' twiddles in the order it stores them (fft.hip.h "first-pass twiddles")
    PCD_HIP_TRY(hipMalloc(&t->tw0_fwd, bytes));
    PCD_HIP_TRY(hipMalloc(&t->tw0_inv, bytes));
    hipLaunchKernelGGL(fft_first_pass_twiddles<FT>, dim3((n + 255) / 256), dim3(256), 0, st, t->tw_fwd, t->tw0_fwd, log_n, plan[0].d);
    hipLaunchKernelGGL(fft_first_pass_twiddles<FT>, dim3((n + 255) / 256), dim3(256), 0, st, t->tw_inv, t->tw0_inv, log_n, plan[0].d);
  }
  PCD_HIP_TRY(hipGetLastError());
  return hipStreamSynchronize(st);
}

// transform x in place (tmp = ping-pong partner).  inverse: w^-1 and 1/n;  coset: see fft.hip.h header.
// batch > 1: `batch` transforms of 2^log_n elements each, laid out back to back in x (and in tmp), in one launch per pass
hipError_t run_batched(hipStream_t st, const FftTables& t, uint32_t* x, uint32_t* tmp, int log_n, int inverse, int coset,
                       float* pass_ms, int* npasses, uint32_t batch) {
  std::vector<FftPass> plan = fft_plan(log_n);
  const int P = (int)plan.size();
  const uint32_t* tw = (inverse & 3) ? t.tw_inv : t.tw_fwd;   // (bit 0: inverse, bit 1: raw inverse root, bit 2: keep -- not a direction)
  DomainConsts c;
  memcpy(&c, t.consts, sizeof c);
  FT scale = FT::one();
  int use_scale = 0;
  const bool raw = (inverse & 2) != 0;  // inverse root, no 1/n (a row transform inside a mixed-radix domain)
  const bool keep = (inverse & 4) != 0; // leave the result where the last pass wrote it (tmp after an odd number of passes): the caller
                                        // chains a second transform from there -- ifft then coset_fft of the witness map: no copy at all
  inverse &= 1;
  if (inverse && !coset && !raw) {  // plain 1/n: constant multiply in the last pass
    scale = c.ninv;
    use_scale = 1;
  }
  std::vector<hipEvent_t> ev;
  if (pass_ms) { ev.resize(P + 1); for (auto& e : ev) PCD_HIP_TRY(hipEventCreate(&e)); PCD_HIP_TRY(hipEventRecord(ev[0], st)); }
  uint32_t* src = x;
  uint32_t* dst = tmp;
  int logs = 0;
  for (int i = 0; i < P; i++) {
    const int d = plan[i].d, logT = plan[i].logT;
    const bool first = (i == 0), last = (i == P - 1);
    const uint32_t* pre = (first && coset && !inverse) ? t.coset : nullptr;
    const uint32_t* post = (last && coset && inverse) ? t.coset_inv_scaled : nullptr;
    const uint32_t blocks = 1u << (log_n - d - logT);
    const size_t lds = ((size_t)1 << (d + logT)) * EW * 4;
    FTP scale_p;
    memcpy(&scale_p, &scale, sizeof scale);
    hipLaunchKernelGGL(fft_pass_kernel<FTP>, dim3(blocks, batch), dim3(256), lds, st, src, dst, tw, log_n, d, logT, logs, pre, post,
                       (last ? use_scale : 0), scale_p, ((size_t)EW) << log_n,
                       (first && P > 1) ? ((inverse || raw) ? t.tw0_inv : t.tw0_fwd) : nullptr);
    if (pass_ms) PCD_HIP_TRY(hipEventRecord(ev[i + 1], st));
    logs += d;
    std::swap(src, dst);
  }
  PCD_HIP_TRY(hipGetLastError());
  if (src != x && !keep) PCD_HIP_TRY(hipMemcpyAsync(x, src, ((size_t)batch << log_n) * EW * 4, hipMemcpyDeviceToDevice, st));
  if (pass_ms) {
    PCD_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < P && i < 8; i++) (void)hipEventElapsedTime(&pass_ms[i], ev[i], ev[i + 1]);
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  if (npasses) *npasses = P;
  return hipSuccess;
}
hipError_t run(hipStream_t st, const FftTables& t, uint32_t* x, uint32_t* tmp, int log_n, int inverse, int coset, float* pass_ms, int* npasses) {
  return run_batched(st, t, x, tmp, log_n, inverse, coset, pass_ms, npasses, 1);
}
hipError_t run_batched_entry(hipStream_t st, const FftTables& t, uint32_t* x, uint32_t* tmp, int log_n, int inverse, int coset, int* npasses,
                             uint32_t batch) {
  return run_batched(st, t, x, tmp, log_n, inverse, coset, nullptr, npasses, batch);
}

// ---- mixed-radix domain n = m * 2^a
struct MixedConsts { FT wn, wn_inv, wm, wm_inv, ninv, zinv; };
__global__ void mixed_consts_kernel(uint32_t n, uint32_t m, uint32_t* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  typedef typename FT::Params P;
  // e = (p - 1) / n: p - 1 = (p - 2) + 1 on 32-bit words, then long division
  constexpr int NW = P::N32;
  uint32_t pm1[NW], e[NW];
  uint64_t carry = 1;
  for (int i = 0; i < NW; i++) { uint64_t x = (uint64_t)P::modm2(i) + carry; pm1[i] = (uint32_t)x; carry = x >> 32; }
  uint64_t rem = 0;
  for (int i = NW - 1; i >= 0; i--) { uint64_t cur = (rem << 32) | pm1[i]; e[i] = (uint32_t)(cur / n); rem = cur % n; }
  FT g = FT::generator(), w = FT::one();
  bool started = false;
  for (int i = NW * 32 - 1; i >= 0; i--) {
    if (started) w = w.sqr();
    if ((e[i >> 5] >> (i & 31)) & 1) { w = started ? w * g : g; started = true; }
  }
  FT wm = w.pow_u64(n / m);
  FT zinv = (g.pow_u64(n) - FT::one()).inv();
  w.store(out);
  w.inv().store(out + EW);
  wm.store(out + 2 * EW);
  wm.inv().store(out + 3 * EW);
  FT::from_u64(n).inv().store(out + 4 * EW);
  zinv.store(out + 5 * EW);
}
hipError_t mixed_make_tables(hipStream_t st, uint32_t n, uint32_t m, FftTables* t) {
  if (rem_check(n, m)) return hipErrorInvalidValue;
  uint32_t* d = nullptr;
  PCD_HIP_TRY(hipMalloc(&d, 6 * EW * 4));
  hipLaunchKernelGGL(mixed_consts_kernel, dim3(1), dim3(64), 0, st, n, m, d);
  MixedConsts c;
  hipError_t e = hipMemcpyAsync(&c, d, 6 * EW * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  PCD_HIP_TRY(e);
  static_assert(sizeof(MixedConsts) <= sizeof(t->consts), "consts buffer too small");
  memcpy(t->consts, &c, sizeof c);
  const uint32_t N2 = n / m;
  PCD_HIP_TRY(hipMalloc(&t->tw_fwd, (size_t)N2 * EW * 4));
  PCD_HIP_TRY(hipMalloc(&t->tw_inv, (size_t)N2 * EW * 4));
  PCD_HIP_TRY(hipMalloc(&t->coset, (size_t)n * EW * 4));
  PCD_HIP_TRY(hipMalloc(&t->coset_inv_scaled, (size_t)n * EW * 4));
  FT g = FT::generator();
  DomainConsts dc;  // g^-1 via the radix-2 helper (log_n irrelevant for g)
  PCD_HIP_TRY(get_consts(st, 1, &dc));
  dim3 bd(64);
  hipLaunchKernelGGL(fft_fill_powers<FT>, dim3(((N2 + 255) / 256 + 63) / 64), bd, 0, st, t->tw_fwd, N2, c.wn, FT::one());
  hipLaunchKernelGGL(fft_fill_powers<FT>, dim3(((N2 + 255) / 256 + 63) / 64), bd, 0, st, t->tw_inv, N2, c.wn_inv, FT::one());
  hipLaunchKernelGGL(fft_fill_powers<FT>, dim3(((n + 255) / 256 + 63) / 64), bd, 0, st, t->coset, n, g, FT::one());
  hipLaunchKernelGGL(fft_fill_powers<FT>, dim3(((n + 255) / 256 + 63) / 64), bd, 0, st, t->coset_inv_scaled, n, dc.ginv, c.ninv);
  PCD_HIP_TRY(hipGetLastError());
  return hipStreamSynchronize(st);
}
// x (n elements) transformed in place; tmp: n elements; t2: radix-2 tables of the 2^a row domain
hipError_t mixed_run(hipStream_t st, const FftTables& t, const FftTables& t2, uint32_t* x, uint32_t* tmp, uint32_t m, int a,
                     int inverse, int coset) {
  MixedConsts c;
  memcpy(&c, t.consts, sizeof c);
  const uint32_t N2 = 1u << a, n = N2 * m;
  const uint32_t* pre = (coset && !inverse) ? t.coset : nullptr;
  hipLaunchKernelGGL(fft_mixed_columns_kernel<FT>, dim3((N2 + 63) / 64), dim3(64), 0, st, x, tmp, inverse ? t.tw_inv : t.tw_fwd, N2, m,
                     inverse ? c.wm_inv : c.wm, pre);
  // the m rows through the radix-2 passes, all rows in one launch per pass; x serves as their ping-pong space
  PCD_HIP_TRY(run_batched(st, t2, tmp, x, a, inverse ? 3 : 0, 0, nullptr, nullptr, m));
  const uint32_t* post = (coset && inverse) ? t.coset_inv_scaled : nullptr;
  const int use_scale = (inverse && !coset) ? 1 : 0;
  hipLaunchKernelGGL(fft_mixed_interleave_kernel<FT>, dim3((n + 255) / 256), dim3(256), 0, st, tmp, x, N2, m, post, use_scale, c.ninv);
  return hipGetLastError();
}
hipError_t mixed_mul_sub_divz(hipStream_t st, const FftTables& t, uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t n) {
  MixedConsts mc;
  memcpy(&mc, t.consts, sizeof mc);
  hipLaunchKernelGGL(fft_mul_sub_scale<FT>, dim3((n + 255) / 256), dim3(256), 0, st, a, b, c, n, mc.zinv);
  return hipGetLastError();
}

// out[i] = canonical words of in[i] * k (in: device image, k: C-ABI Montgomery element on the device).  first_is_one:
// element 0 is taken as 1 (the leading one of an R1CS assignment, whatever the caller stored there).
__global__ void __launch_bounds__(256) scale_canon_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ k_abi,
                                                          uint32_t* __restrict__ out, uint32_t n, int has_k, int first_is_one) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FT v = (i == 0 && first_is_one) ? FT::one() : FT::load(in + (size_t)i * EW);
  if (has_k) v = v * FT::from_abi(k_abi);
  v.to_canonical_words(out + (size_t)i * FT::ABI_WORDS);
}
hipError_t scale_canon(hipStream_t st, const uint32_t* in, const uint32_t* k_abi, uint32_t* out, uint32_t n, int first_is_one) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scale_canon_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in, k_abi, out, n, k_abi ? 1 : 0, first_is_one);
  return hipGetLastError();
}

hipError_t convert(hipStream_t st, const uint32_t* in, uint32_t* out, uint32_t n, int mode) {
  if (n == 0) return hipSuccess;
  dim3 gd((n + 255) / 256), bd(256);
  switch (mode) {
    case 0: hipLaunchKernelGGL((fp_convert_kernel<FT, 0>), gd, bd, 0, st, in, out, n); break;
    case 1: hipLaunchKernelGGL((fp_convert_kernel<FT, 1>), gd, bd, 0, st, in, out, n); break;
    case 2: hipLaunchKernelGGL((fp_convert_kernel<FT, 2>), gd, bd, 0, st, in, out, n); break;
    case 3: hipLaunchKernelGGL((fp_convert_kernel<FT, 3>), gd, bd, 0, st, in, out, n); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- the three mat-vecs of the witness map: out[i] = <M_i, z> (i < rows);  out[rows + j] = z[j] (j < num_inputs) if append_inputs;
// 0 up to n.  Entries with small integer coefficients (DevCsr, common.h) cost additions, not products: a lane sums c * z[col] limb-wise
// in 64 bits for up to SPMV_FLUSH entries (sum |c| <= 2000) and reduces once (Fp::from_signed_sum); only the heavy entries take a
// Montgomery product.  The two kinds are separate loops, so a wave whose rows hold no heavy entry never executes a product.
constexpr int SPMV_FLUSH = 48;  // light entries per reduction: 48 x 32 <= 2000
struct SpmvLight {
  int64_t s[FT::N];
  int cnt = 0;
  PCD_DEV SpmvLight() {
#pragma unroll
    for (int i = 0; i < FT::N; i++) s[i] = 0;
  }
  PCD_DEV void add(int c, const FT& v, FT& acc) {
#pragma unroll
    for (int i = 0; i < FT::N; i++) s[i] += (int64_t)c * (int64_t)(int32_t)v.v[i];   // (limbs below 2^29: one v_mad_i64_i32)
    if (++cnt == SPMV_FLUSH) flush(acc);
  }
  PCD_DEV void flush(FT& acc) {
    if (!cnt) return;
    acc = acc + FT::from_signed_sum(s);
#pragma unroll
    for (int i = 0; i < FT::N; i++) s[i] = 0;
    cnt = 0;
  }
};
// one lane per row (rows of up to SPMV_LONG_ROW entries; longer ones are left to the long-row kernel), the tail of the vector too
PCD_DEV void spmv_row(const DevCsr& m, const uint32_t* __restrict__ z, uint32_t num_inputs, int append_inputs, uint32_t n, uint32_t* __restrict__ out,
                      uint32_t i) {
  if (i >= n) return;
  FT acc = FT::zero();
  if (i < m.rows) {
    const uint64_t lo = m.rp[i], hi = m.rp[i + 1];
    if (hi - lo > SPMV_LONG_ROW) return;
    const uint64_t mid = lo + m.nl[i];
    SpmvLight L;
    for (uint64_t k = lo; k < mid; k++) L.add((int)m.lc[k], FT::load(z + (size_t)m.col[k] * EW), acc);
    L.flush(acc);
    for (uint64_t k = mid; k < hi; k++) acc = acc + FT::load(m.coeff + k * EW) * FT::load(z + (size_t)m.col[k] * EW);
  } else if (append_inputs && i < m.rows + num_inputs) {
    acc = FT::load(z + (size_t)(i - m.rows) * EW);
  }
  acc.store(out + (size_t)i * EW);
}
// one wave per long row: lanes stride over the entries, then a butterfly of field additions over the wave
PCD_DEV void spmv_long_row(const DevCsr& m, const uint32_t* __restrict__ z, uint32_t* __restrict__ out, uint32_t w, uint32_t lane) {
  if (w >= m.n_long) return;
  const uint32_t i = m.long_rows[w];
  const uint64_t lo = m.rp[i], hi = m.rp[i + 1], mid = lo + m.nl[i];
  FT acc = FT::zero();
  SpmvLight L;
  for (uint64_t k = lo + lane; k < mid; k += 64) L.add((int)m.lc[k], FT::load(z + (size_t)m.col[k] * EW), acc);
  L.flush(acc);
  for (uint64_t k = mid + lane; k < hi; k += 64) acc = acc + FT::load(m.coeff + k * EW) * FT::load(z + (size_t)m.col[k] * EW);
  for (int d = 32; d > 0; d >>= 1) {
    FT o;
#pragma unroll
    for (int q = 0; q < FT::N; q++) o.v[q] = (uint32_t)__shfl_xor((int)acc.v[q], d, 64);
    acc = acc + o;
  }
  if (lane == 0) acc.store(out + (size_t)i * EW);
}
__global__ void __launch_bounds__(256) spmv_kernel(const DevCsr m, const uint32_t* __restrict__ z, uint32_t num_inputs, int append_inputs,
                                                   uint32_t n, uint32_t* __restrict__ out) {
  spmv_row(m, z, num_inputs, append_inputs, n, out, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(64) spmv_long_kernel(const DevCsr m, const uint32_t* __restrict__ z, uint32_t* __restrict__ out) {
  spmv_long_row(m, z, out, blockIdx.x, threadIdx.x);
}
// the three mat-vecs of a witness map in one launch each (blockIdx.y = the matrix; outputs `stride` words apart; matrix 0 appends the inputs)
struct DevCsr3 { DevCsr m[3]; };
__global__ void __launch_bounds__(256) spmv3_kernel(const DevCsr3 mm, const uint32_t* __restrict__ z, uint32_t num_inputs, uint32_t n,
                                                    uint32_t* __restrict__ out, size_t stride) {
  spmv_row(mm.m[blockIdx.y], z, num_inputs, blockIdx.y == 0 ? 1 : 0, n, out + blockIdx.y * stride, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(64) spmv3_long_kernel(const DevCsr3 mm, const uint32_t* __restrict__ z, uint32_t* __restrict__ out, size_t stride) {
  spmv_long_row(mm.m[blockIdx.y], z, out + blockIdx.y * stride, blockIdx.x, threadIdx.x);
}
hipError_t spmv(hipStream_t st, const DevCsr& m, const uint32_t* z, uint32_t num_inputs, int append_inputs, uint32_t n, uint32_t* out) {
  hipLaunchKernelGGL(spmv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, m, z, num_inputs, append_inputs, n, out);
  if (m.n_long) hipLaunchKernelGGL(spmv_long_kernel, dim3(m.n_long), dim3(64), 0, st, m, z, out);
  return hipGetLastError();
}
hipError_t spmv3(hipStream_t st, const DevCsr mats[3], const uint32_t* z, uint32_t num_inputs, uint32_t n, uint32_t* out, size_t stride_words) {
  DevCsr3 mm;
  uint32_t max_long = 0;
  for (int k = 0; k < 3; k++) { mm.m[k] = mats[k]; max_long = std::max(max_long, mats[k].n_long); }
  hipLaunchKernelGGL(spmv3_kernel, dim3((n + 255) / 256, 3), dim3(256), 0, st, mm, z, num_inputs, n, out, stride_words);
  if (max_long) hipLaunchKernelGGL(spmv3_long_kernel, dim3(max_long, 3), dim3(64), 0, st, mm, z, out, stride_words);
  return hipGetLastError();
}
// host: the C-ABI image of a small integer (the field templates are __host__ __device__)
void small_abi(int c, uint32_t* out) {
  FT v = FT::from_u64((uint64_t)(c < 0 ? -c : c));
  if (c < 0) v = v.neg();
  v.to_abi(out);
}

hipError_t mul_sub_divz(hipStream_t st, const FftTables& t, uint32_t* a, const uint32_t* b, const uint32_t* c, int log_n) {
  DomainConsts dc;
  memcpy(&dc, t.consts, sizeof dc);
  const uint32_t n = 1u << log_n;
  hipLaunchKernelGGL(fft_mul_sub_scale<FT>, dim3((n + 255) / 256), dim3(256), 0, st, a, b, c, n, dc.zinv);
  return hipGetLastError();
}

// ---- Groth16 generator scalars (ark-groth16 `generate_parameters` -> `R1CSToQAP::instance_map_with_evaluation`, reached
// from the reference through circuit_specific_setup, src/ec_cycle_pcd/mod.rs:69,78): SURVEY.md 8(f) rank 2.
// toxic = [alpha, beta, gamma, delta, tau] (C-ABI Montgomery, on the device).
// consts (device image): [0] Z(tau)/n  [1] 1/delta  [2] 1/gamma  [3] Z(tau)/delta  [4] alpha  [5] beta  [6] tau
constexpr int SETUP_CONSTS = 7;
__global__ void setup_consts_kernel(const uint32_t* __restrict__ toxic, uint32_t n, FT ninv, uint32_t* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  constexpr int AW = FT::ABI_WORDS;
  const FT alpha = FT::from_abi(toxic), beta = FT::from_abi(toxic + AW), gamma = FT::from_abi(toxic + 2 * AW);
  const FT delta = FT::from_abi(toxic + 3 * AW), tau = FT::from_abi(toxic + 4 * AW);
  const FT zt = tau.pow_u64(n) - FT::one();
  const FT dinv = delta.inv();
  (zt * ninv).store(out);
  dinv.store(out + EW);
  gamma.inv().store(out + 2 * EW);
  (zt * dinv).store(out + 3 * EW);
  alpha.store(out + 4 * EW);
  beta.store(out + 5 * EW);
  tau.store(out + 6 * EW);
}
// u[i] = L_i(tau) = Z(tau) w^i / (n (tau - w^i)): `evaluate_all_lagrange_coefficients`; LAGRANGE_BATCH consecutive i per
// lane share one inversion.  tau inside the domain (upstream samples it outside) raises *err.
constexpr int LAGRANGE_BATCH = 8;
__global__ void __launch_bounds__(256) setup_lagrange_kernel(const uint32_t* __restrict__ consts, FT w, uint32_t n, uint32_t* __restrict__ u,
                                                             uint32_t* __restrict__ err) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lo = (uint64_t)lane * LAGRANGE_BATCH;
  if (lo >= n) return;
  const int cnt = (int)((n - lo) < (uint64_t)LAGRANGE_BATCH ? (n - lo) : (uint64_t)LAGRANGE_BATCH);
  const FT scale = FT::load(consts), tau = FT::load(consts + 6 * EW);
  FT pre[LAGRANGE_BATCH], den[LAGRANGE_BATCH];
  FT wi = w.pow_u64(lo), run = FT::one();
  const FT w0 = wi;
  for (int k = 0; k < cnt; k++) {
    den[k] = tau - wi;
    if (den[k].is_zero()) { atomicOr(err, 1u); den[k] = FT::one(); }
    pre[k] = run;
    run = run * den[k];
    wi = wi * w;
  }
  FT inv = run.inv();
  // w^(lo + k) again, from the top: wi / w = w^(lo + cnt - 1)
  FT wk[LAGRANGE_BATCH];
  wi = w0;
  for (int k = 0; k < cnt; k++) { wk[k] = wi; wi = wi * w; }
  for (int k = cnt - 1; k >= 0; k--) {
    const FT di = inv * pre[k];
    inv = inv * den[k];
    (scale * wk[k] * di).store(u + (lo + k) * EW);
  }
}
// per variable i < m:  a = At_i (+ u[nc + i] for the instance variables), b = Bt_i, t = (beta a + alpha b + Ct_i) / gamma (instance)
// or / delta (witness);  per i < h_len:  h = Z(tau)/delta * tau^i.  Canonical words out; b goes to both scalar arrays.
__global__ void __launch_bounds__(256) setup_scalars_kernel(const uint32_t* __restrict__ at, const uint32_t* __restrict__ bt,
                                                            const uint32_t* __restrict__ ct, const uint32_t* __restrict__ u,
                                                            const uint32_t* __restrict__ consts, uint32_t nc, uint32_t m, uint32_t ni,
                                                            uint32_t h_len, uint32_t* __restrict__ a_can, uint32_t* __restrict__ b_can,
                                                            uint32_t* __restrict__ t_can, uint32_t* __restrict__ h_can,
                                                            uint32_t* __restrict__ b2_can) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int AW = FT::ABI_WORDS;
  if (i < m) {
    FT a = FT::load(at + (size_t)i * EW);
    if (i < ni) a = a + FT::load(u + (size_t)(nc + i) * EW);
    const FT b = FT::load(bt + (size_t)i * EW), c = FT::load(ct + (size_t)i * EW);
    const FT alpha = FT::load(consts + 4 * EW), beta = FT::load(consts + 5 * EW);
    FT t = beta * a + alpha * b + c;
    t = t * FT::load(consts + (i < ni ? 2 : 1) * EW);
    a.to_canonical_words(a_can + (size_t)i * AW);
    b.to_canonical_words(b_can + (size_t)i * AW);
    b.to_canonical_words(b2_can + (size_t)i * AW);
    t.to_canonical_words(t_can + (size_t)i * AW);
  }
  if (i < h_len) {
    const FT tau = FT::load(consts + 6 * EW);
    (FT::load(consts + 3 * EW) * tau.pow_u64(i)).to_canonical_words(h_can + (size_t)i * AW);
  }
}
hipError_t setup_scalars(hipStream_t st, const void* domain_consts, const uint32_t* toxic_abi, uint32_t n, uint32_t nc, uint32_t m,
                         uint32_t ni, const uint32_t* at, const uint32_t* bt, const uint32_t* ct, uint32_t* u, uint32_t* consts_dev,
                         uint32_t* err_dev, uint32_t* a_can, uint32_t* b_can, uint32_t* t_can, uint32_t* h_can, uint32_t* b2_can, int phase) {
  // both DomainConsts and MixedConsts start with the domain generator and keep 1/n in slot 4
  FT w, ninv;
  memcpy(&w, domain_consts, sizeof w);
  memcpy(&ninv, (const char*)domain_consts + 4 * sizeof(FT), sizeof ninv);
  if (phase == 0) {  // u = Lagrange coefficients at tau (needed by the transposed mat-vecs)
    hipLaunchKernelGGL(setup_consts_kernel, dim3(1), dim3(64), 0, st, toxic_abi, n, ninv, consts_dev);
    const uint32_t lanes = (n + LAGRANGE_BATCH - 1) / LAGRANGE_BATCH;
    hipLaunchKernelGGL(setup_lagrange_kernel, dim3((lanes + 255) / 256), dim3(256), 0, st, consts_dev, w, n, u, err_dev);
  } else {
    const uint32_t h_len = n - 1, cnt = m > h_len ? m : h_len;
    hipLaunchKernelGGL(setup_scalars_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, at, bt, ct, u, consts_dev, nc, m, ni, h_len, a_can,
                       b_can, t_can, h_can, b2_can);
  }
  return hipGetLastError();
}

}  // namespace

#define PCD_CAT_(a, b) a##b
#define PCD_CAT(a, b) PCD_CAT_(a, b)
const FieldEntry* PCD_CAT(pcd_field_entry_, PCD_FIELD_IDX)() {
  static const FieldEntry e = {EW, FT::ABI_WORDS, FT::Params::TWO_ADICITY, make_tables, run, convert, spmv, small_abi, mul_sub_divz,
                               mixed_make_tables, mixed_run, mixed_mul_sub_divz, scale_canon, SETUP_CONSTS, setup_scalars, run_batched_entry, spmv3};
  return &e;
}

}  // namespace pcd
