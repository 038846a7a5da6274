// Variable-base multi-scalar multiplication on gfx950 (K3/K4 of SURVEY.md section 8).
//
// Replaces ark-ec `VariableBaseMSM::multi_scalar_mul` [upstream msm/variable_base.rs] as called by
// Groth16's prover under `IC::MainSNARK::prove` / `IC::HelpSNARK::prove`
// (/root/reference src/ec_cycle_pcd/mod.rs:171,179): sum_i k_i * P_i for affine bases P_i and
// canonical scalars k_i.  The value is a unique group element, so the GPU is free to organise the
// bucket method differently from upstream (which parallelises over windows only):
//
//   1. sort          signed digits of every scalar -> (bucket, base index) entries sorted by bucket: an MSD partition through LDS
//                    (msm_coarse_kernel x2, msm_bin_scan_kernel, msm_bin_sort_kernel); counting-sort and single-pass binning
//                    variants stay selectable (msm_digits_kernel, scan_*)
//   2. msm_accumulate fixed-size chunks of the sorted list, one lane (or 2 / 3 lanes: lane-split extension fields) per chunk:
//                    every lane does the same number of mixed additions whatever the scalar distribution; runs that cross
//                    a chunk edge are emitted as "pieces"; lazily reduced XYZZ accumulator for the 298-bit G1 (int-VALU-bound)
//   3. msm_fixup / msm_big_segments / msm_big_bucket   pieces of one bucket are summed; flushed records become Jacobian points
//   4. msm_tail_level / msm_tail_pair  sum_d d*B_d by blocked running sums, then pair levels (two lanes per group operation)
//   5. msm_horner    windows combined by c doublings each (not needed with one precomputed copy of the bases per window)
//
// Zero digits never enter the sorted list (upstream skips zero scalars); scalars equal to one go, staged per workgroup, to a list
// of their own that forms a pseudo bucket added to bucket (window 0, digit 1), so that bit-decomposition-heavy witnesses neither
// serialise on one counter nor pile up in one chunk-straddling run of the ordinary list.
#pragma once
#include <cstdlib>
#include <math.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "ec.hip.h"

namespace pcd {

// a few HIP events that are destroyed on every exit path
template <int K>
struct EventSet {
  hipEvent_t e[K] = {};
  bool live = false;
  hipError_t create() {
    for (int i = 0; i < K; i++) { hipError_t r = hipEventCreate(&e[i]); if (r != hipSuccess) return r; }
    live = true;
    return hipSuccess;
  }
  hipEvent_t operator[](int i) const { return e[i]; }
  ~EventSet() { for (int i = 0; i < K; i++) if (e[i]) (void)hipEventDestroy(e[i]); }
};

struct MsmPlan {
  uint32_t n = 0;
  int c = 0;           // window bits
  int W = 0;           // windows
  uint32_t nkeys = 0;  // W << c
  uint32_t chunk = 0;  // sorted entries per lane in msm_accumulate
};

// windows of a signed-digit decomposition (see "digits" below): ceil((bits + 1) / c), room for the last carry
PCD_HD constexpr int msm_num_windows(int scalar_bits, int c) { return (scalar_bits + c) / c; }

// minimise  W*n mixed adds (11 M) + Wg * 2^(c+1) full adds (16 M) over c, where Wg = bucket windows:
// Wg = W without precomputed bases, Wg = ceil(W / groups) with them (full_precompute: Wg = 1).
inline int msm_pick_window(size_t n, int scalar_bits, int full_precompute) {
  double best = 1e300;
  int bc = 8;
  for (int c = 6; c <= 22; c++) {
    int W = msm_num_windows(scalar_bits, c);
    int Wg = full_precompute ? 1 : W;
    double cost = (double)W * (double)n * 11.0 + (double)Wg * (double)(1u << c) * 16.0 * 1.3;  // 2^(c-1) buckets per window
    // a short top window concentrates n entries on 2^top buckets: their histogram / scatter atomics serialise
    // (measured: ~35 ns per entry per hot bucket, i.e. ~1400 modmul-times)
    int top = scalar_bits + 1 - (W - 1) * c;
    cost += (double)n / (double)(1u << (top > 20 ? 20 : top)) * 1400.0;
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

// ------------------------------------------------------------------------------------------------ digits
// Signed digits: k = sum_w d_w 2^(c w) with d_w in [-2^(c-1), 2^(c-1)]: half as many buckets per window (bucket |d|, the
// point negated when d < 0), i.e. one more bit per window for the same bucket array.  W = ceil((bits + 1) / c) windows leave
// room for the last carry.  An entry = base index | sign << 31.
constexpr uint32_t MSM_NEG = 0x80000000u;
// raw digit + carry in -> |d| (0 .. 2^(c-1)), sign flag (MSM_NEG or 0), carry out
PCD_DEV uint32_t msm_signed(uint32_t raw, int c, uint32_t& carry, uint32_t& neg) {
  uint32_t d = raw + carry;
  const bool n = d > (1u << (c - 1));
  carry = n ? 1u : 0u;
  neg = n ? MSM_NEG : 0u;
  return n ? (1u << c) - d : d;
}
// a scalar must be a reduced canonical value (< r < 2^bits): the signed recoding leaves room for one carry only, so anything
// wider would silently drop its top bits.  Offenders raise a device flag that the host-result entry points turn into PCDHIP_E_ARG.
template <int NS>
PCD_DEV bool msm_scalar_too_wide(const uint32_t* s, int bits) {
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    if (32 * k >= bits) o |= s[k];
    else if (32 * (k + 1) > bits) o |= s[k] >> (bits - 32 * k);
  }
  return o != 0;
}
// Bases that are the point at infinity contribute nothing whatever their scalar: with a bitmap of them (one bit per base of the resident
// vector; null when it holds none) their entries never enter the bucket lists.  The queries of a Groth16 key are full of them -- a
// variable that no constraint's A (B) row mentions has a_query (b_query) = O: 28 % / 41 % of the variables of the bench's R1CS -- and a
// lane that meets one in its chunk idles for a whole mixed addition while its wave works.
PCD_DEV bool msm_base_is_inf(const uint32_t* __restrict__ inf_bits, uint32_t idx) { return inf_bits && ((inf_bits[idx >> 5] >> (idx & 31u)) & 1u); }
template <int NS>
PCD_DEV uint32_t msm_digit(const uint32_t* s, int w, int c) {
  int bit = w * c;
  int word = bit >> 5, off = bit & 31;
  uint64_t x = s[word];
  if (word + 1 < NS) x |= (uint64_t)s[word + 1] << 32;
  return (uint32_t)(x >> off) & ((1u << c) - 1u);
}

// Window w of the scalar belongs to group g = w / Wg and bucket window j = w % Wg: with precomputed
// bases the point used is 2^(c Wg g) P_i, stored at index g * n_total + i, so all groups share the
// same Wg bucket windows (Wg = 1: no window combine at all).
//
// Sorting of the (bucket, base index) entries.  Keys 0 .. nkeys-1 are the real buckets, key `nkeys` is a pseudo
// bucket for the scalars equal to one (their entries are appended wave-aggregated to `ones_idx`; the pseudo bucket
// is added to bucket (window 0, digit 1) before the reduction).
//   MODE_BIN     one pass: entry -> slots[key][pos] with pos = atomicAdd(cnt[key]) while pos < cap; cnt ends as the
//                exact histogram; `*flag` is raised if any bucket overflowed its cap slots
//   MODE_SCATTER compact counting-sort scatter through `off` (cursor array `cnt`); with `flag` given it runs only
//                when the flag is raised (the fallback of MODE_BIN) and leaves the ones alone
//   MODE_HIST    histogram only (first pass of the two-pass sort used when slots are not affordable)
enum { MODE_HIST = 0, MODE_SCATTER = 1, MODE_BIN = 2 };
template <int NS, int MODE>
__global__ void __launch_bounds__(256) msm_digits_kernel(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, int Wg,
                                                         uint32_t n_total, uint32_t base_offset, uint32_t nkeys,
                                                         uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                                         uint32_t* __restrict__ sorted_idx, uint32_t* __restrict__ slots, uint32_t cap,
                                                         uint32_t* __restrict__ ones_idx, uint32_t* __restrict__ flag, int skip_ones,
                                                         int scalar_bits, uint32_t* __restrict__ err, const uint32_t* __restrict__ inf_bits) {
  if (MODE == MODE_SCATTER && flag && *flag == 0) return;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = i < n && !msm_base_is_inf(inf_bits, base_offset + i);
  uint32_t s[NS];
  bool is_one = false;
  // (the width check looks at EVERY scalar, also those whose base is the point at infinity and so never enter a list: an unreduced
  //  scalar is a caller's error wherever it sits -- include/pcdhip.h promises PCDHIP_E_ARG for it)
  if (live || (i < n && MODE != MODE_SCATTER && err)) {
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) { s[k] = scalars[(size_t)i * NS + k]; if (k) hi |= s[k]; }
    is_one = live && (hi == 0 && s[0] == 1);
    if (MODE != MODE_SCATTER && err && msm_scalar_too_wide<NS>(s, scalar_bits)) *err = 1u;
  }
  // scalars equal to one: one atomic per wave on the pseudo bucket
  unsigned long long m = __ballot(live && is_one);
  if (m && !skip_ones) {
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&cnt[nkeys], (uint32_t)__popcll(m));
    if (MODE != MODE_HIST) {
      base = __shfl(base, leader, 64);
      if (live && is_one) {
        uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (MODE == MODE_BIN) ones_idx[base + rank] = base_offset + i;
        else sorted_idx[off[nkeys] + base + rank] = base_offset + i;
      }
    }
  }
  if (!live || is_one) return;
  int g = 0, j = 0;
  uint32_t carry = 0, neg;
  for (int w = 0; w < W; w++) {
    uint32_t d = msm_signed(msm_digit<NS>(s, w, c), c, carry, neg);
    if (d != 0) {
      uint32_t key = ((uint32_t)j << c) | d;
      uint32_t pos = atomicAdd(&cnt[key], 1u);
      uint32_t idx = ((uint32_t)g * n_total + base_offset + i) | neg;
      if (MODE == MODE_SCATTER) sorted_idx[off[key] + pos] = idx;
      if (MODE == MODE_BIN) {
        if (pos < cap) slots[(size_t)key * cap + pos] = idx;
        else *flag = 1u;
      }
    }
    if (++j == Wg) { j = 0; g++; }
  }
}

// ------------------------------------------------------------------------------------------------ partition sort
// Default sort of the (bucket, base) entries: an MSD partition through LDS instead of one global atomic per entry.
//   bins of 2^BIN_SHIFT consecutive keys;
//   pass 0  msm_coarse_kernel<false>: per-workgroup LDS histogram over the bins of a tile of scalars, one global atomic
//           per (workgroup, non-empty bin);  scalars equal to one go, wave-aggregated, to the pseudo bucket's list
//   scan    bin bases
//   pass 1  msm_coarse_kernel<true>: the same LDS histogram, one global atomic per (workgroup, bin) to reserve a run in
//           the bin, then the entries (key, base index) are written into that run through LDS cursors
//   pass 2  msm_bin_sort_kernel: one workgroup per bin: LDS counting sort of its entries by the low key bits; writes
//           the final base-index list and the per-key offsets (bins are in key order, so no global scan is needed)
// Global atomics drop from n W to ~2 n W / TILE_ENTRIES_PER_BIN; everything else is LDS atomics and streaming.
#ifndef PCD_SORT_UNROLL
#define PCD_SORT_UNROLL 4
#endif
constexpr int MSM_SORT_UNROLL = PCD_SORT_UNROLL;  // entries per lane in flight in the per-bin counting sort
constexpr int MSM_BIN_SHIFT = 9;          // 512 keys per bin
#ifndef PCD_MSM_TILE
#define PCD_MSM_TILE 2048
#endif
constexpr int MSM_TILE = PCD_MSM_TILE;    // scalars per workgroup in passes 0 / 1
constexpr uint32_t MSM_MAX_BINS = 8192;   // LDS histogram of a workgroup: 32 KiB

template <int NS, bool WRITE>
__global__ void __launch_bounds__(256) msm_coarse_kernel(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, int Wg,
                                                         uint32_t n_total, uint32_t base_offset, uint32_t nbins,
                                                         uint32_t* __restrict__ gbin /* WRITE: cursors (start at the bin bases) */,
                                                         uint64_t* __restrict__ entries, uint32_t* __restrict__ ones_count,
                                                         uint32_t* __restrict__ ones_idx, int scalar_bits, uint32_t* __restrict__ err,
                                                         const uint32_t* __restrict__ inf_bits) {
  extern __shared__ uint32_t lbin[];  // nbins counters, then (WRITE) reused as cursors
  __shared__ uint32_t s_ones[WRITE ? 1 : MSM_TILE];  // pass 0: base indices of this tile's scalars equal to one
  __shared__ uint32_t s_nones, s_ones_base;
  for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) lbin[b] = 0;
  if (threadIdx.x == 0) s_nones = 0;
  __syncthreads();
  const uint32_t tile0 = blockIdx.x * MSM_TILE;
  // sweep A: histogram of this tile
  for (uint32_t k = threadIdx.x; k < MSM_TILE; k += blockDim.x) {
    uint32_t i = tile0 + k;
    bool live = i < n && !msm_base_is_inf(inf_bits, base_offset + i);
    uint32_t s[NS];
    bool is_one = false;
    if (live || (i < n && !WRITE)) {   // (pass 0 checks the width of every scalar, those on bases at infinity included)
      uint32_t hi = 0;
#pragma unroll
      for (int q = 0; q < NS; q++) { s[q] = scalars[(size_t)i * NS + q]; if (q) hi |= s[q]; }
      is_one = live && (hi == 0 && s[0] == 1);
      if (!WRITE && msm_scalar_too_wide<NS>(s, scalar_bits)) *err = 1u;
    }
    if (!WRITE) {  // the ones are listed once, in pass 0: staged in LDS, ONE global atomic per workgroup (a bit-heavy witness is a
                   // third ones: one atomic per wave on the single counter serialised 16 384 of them, 0.2 ms at n = 2^20)
      unsigned long long m = __ballot(live && is_one);
      if (m) {
        int lane = threadIdx.x & 63;
        int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&s_nones, (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        if (live && is_one) s_ones[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = base_offset + i;
      }
    }
    if (!live || is_one) continue;
    int j = 0;
    uint32_t carry = 0, neg;
    for (int w = 0; w < W; w++) {
      uint32_t d = msm_signed(msm_digit<NS>(s, w, c), c, carry, neg);
      if (d != 0) atomicAdd(&lbin[((((uint32_t)j << c) | d)) >> MSM_BIN_SHIFT], 1u);
      if (++j == Wg) j = 0;
    }
  }
  __syncthreads();
  // one global atomic per non-empty bin: total count (pass 0) or run reservation (pass 1)
  for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) {
    uint32_t cnt = lbin[b];
    if (cnt) {
      uint32_t base = atomicAdd(&gbin[b], cnt);
      if (WRITE) lbin[b] = base;  // becomes this workgroup's cursor inside the bin
    }
  }
  if (!WRITE) {
    if (threadIdx.x == 0) s_ones_base = s_nones ? atomicAdd(ones_count, s_nones) : 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < s_nones; k += blockDim.x) ones_idx[s_ones_base + k] = s_ones[k];
    return;
  }
  __syncthreads();
  // sweep B: write the entries
  for (uint32_t k = threadIdx.x; k < MSM_TILE; k += blockDim.x) {
    uint32_t i = tile0 + k;
    if (i >= n || msm_base_is_inf(inf_bits, base_offset + i)) continue;
    uint32_t s[NS];
    uint32_t hi = 0;
#pragma unroll
    for (int q = 0; q < NS; q++) { s[q] = scalars[(size_t)i * NS + q]; if (q) hi |= s[q]; }
    if (hi == 0 && s[0] <= 1) continue;  // zero or one
    int g = 0, j = 0;
    uint32_t carry = 0, neg;
    for (int w = 0; w < W; w++) {
      uint32_t d = msm_signed(msm_digit<NS>(s, w, c), c, carry, neg);
      if (d != 0) {
        uint32_t key = ((uint32_t)j << c) | d;
        uint32_t pos = atomicAdd(&lbin[key >> MSM_BIN_SHIFT], 1u);
        entries[pos] = ((uint64_t)key << 32) | (uint64_t)(((uint32_t)g * n_total + base_offset + i) | neg);
      }
      if (++j == Wg) { j = 0; g++; }
    }
  }
}

// bin_base: exclusive scan of the bin counts (nbins + 1 values) + the list offsets of the pseudo bucket
static __global__ void __launch_bounds__(1024) msm_bin_scan_kernel(const uint32_t* __restrict__ bin_cnt, uint32_t nbins,
                                                                   uint32_t* __restrict__ bin_base, uint32_t* __restrict__ cursor,
                                                                   const uint32_t* __restrict__ ones_count, uint32_t* __restrict__ off,
                                                                   uint32_t nkeys) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (nbins + 1023) / 1024;
  const uint32_t lo = threadIdx.x * per, hi = min(nbins, lo + per);
  uint32_t acc = 0;
  for (uint32_t b = lo; b < hi; b++) acc += bin_cnt[b];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 1; s < 1024; s <<= 1) {
    uint32_t t = ((int)threadIdx.x >= s) ? part[threadIdx.x - s] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - acc;
  for (uint32_t b = lo; b < hi; b++) { bin_base[b] = run; cursor[b] = run; run += bin_cnt[b]; }
  if (threadIdx.x == 1023) {
    uint32_t M = part[1023];
    bin_base[nbins] = M;
    off[nkeys] = M;                    // pseudo bucket (scalars equal to one) follows the real entries
    off[nkeys + 1] = M + *ones_count;
  }
}

// one workgroup per bin: counting sort of its entries by the low MSM_BIN_SHIFT key bits
static __global__ void __launch_bounds__(256) msm_bin_sort_kernel(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ bin_base,
                                                                  uint32_t* __restrict__ sorted_idx, uint32_t* __restrict__ off) {
  constexpr uint32_t KB = 1u << MSM_BIN_SHIFT;
  __shared__ uint32_t hist[KB];
  __shared__ uint32_t wsum[4];
  const uint32_t b = blockIdx.x, lo = bin_base[b], hi = bin_base[b + 1];
  for (uint32_t k = threadIdx.x; k < KB; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  // (four entries per lane in flight: the loop is a chain of global load -> LDS atomic -> scattered store latencies)
  for (uint32_t p0 = lo + threadIdx.x; p0 < hi; p0 += MSM_SORT_UNROLL * blockDim.x) {
    uint32_t kk[MSM_SORT_UNROLL];
#pragma unroll
    for (int u = 0; u < MSM_SORT_UNROLL; u++) { uint32_t p = p0 + u * blockDim.x; kk[u] = p < hi ? (uint32_t)(entries[p] >> 32) & (KB - 1) : KB; }
#pragma unroll
    for (int u = 0; u < MSM_SORT_UNROLL; u++) if (kk[u] < KB) atomicAdd(&hist[kk[u]], 1u);
  }
  __syncthreads();
  // exclusive scan of the 512 counters: two per lane, wave scan, then the four wave totals
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  uint32_t c0 = hist[2 * t], c1 = hist[2 * t + 1];
  uint32_t incl = c0 + c1;
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += o; }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t q = 0; q < wv; q++) before += wsum[q];
  uint32_t ex = lo + before + incl - (c0 + c1);
  hist[2 * t] = ex;
  hist[2 * t + 1] = ex + c0;
  off[(size_t)b * KB + 2 * t] = ex;
  off[(size_t)b * KB + 2 * t + 1] = ex + c0;
  __syncthreads();
  for (uint32_t p0 = lo + threadIdx.x; p0 < hi; p0 += MSM_SORT_UNROLL * blockDim.x) {
    uint64_t e[MSM_SORT_UNROLL];
    uint32_t pos[MSM_SORT_UNROLL];
#pragma unroll
    for (int u = 0; u < MSM_SORT_UNROLL; u++) { uint32_t p = p0 + u * blockDim.x; e[u] = p < hi ? entries[p] : ~0ull; }
#pragma unroll
    for (int u = 0; u < MSM_SORT_UNROLL; u++) if (e[u] != ~0ull) pos[u] = atomicAdd(&hist[(uint32_t)(e[u] >> 32) & (KB - 1)], 1u);
#pragma unroll
    for (int u = 0; u < MSM_SORT_UNROLL; u++) if (e[u] != ~0ull) sorted_idx[pos[u]] = (uint32_t)e[u];
  }
}

// ------------------------------------------------------------------------------------------------ scan (exclusive, u32)
static __global__ void __launch_bounds__(1024) scan_block_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t per_block,
                                                         uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t sh[1024];
  uint32_t b = blockIdx.x, lo = b * per_block, hi = min(n, lo + per_block);
  uint32_t acc = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) acc += in[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) block_sums[b] = sh[0];
}
static __global__ void __launch_bounds__(1024) scan_apply(const uint32_t* __restrict__ in, uint32_t n, uint32_t per_block,
                                                    const uint32_t* __restrict__ block_sums, uint32_t nblocks,
                                                    uint32_t* __restrict__ out /* n + 1 */) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  uint32_t b = blockIdx.x, lo = b * per_block, hi = min(n, lo + per_block);
  if (threadIdx.x == 0) { uint32_t a = 0; for (uint32_t k = 0; k < b; k++) a += block_sums[k]; carry = a; }
  __syncthreads();
  for (uint32_t base = lo; base < hi; base += 1024) {
    uint32_t i = base + threadIdx.x;
    uint32_t v = (i < hi) ? in[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {  // Hillis-Steele inclusive scan
      uint32_t t = ((int)threadIdx.x >= s) ? sh[threadIdx.x - s] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < hi) out[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += sh[1023];
    __syncthreads();
  }
  if (b == nblocks - 1 && threadIdx.x == 0) out[n] = carry;
}

// ------------------------------------------------------------------------------------------------ accumulate
// Entries per lane of msm_accumulate_kernel, chosen ON THE DEVICE from the actual list length M = off[nkeys] (round 5).  The host only knows
// the largest possible list (n W entries); the real one is shorter by the zero digits, the scalars equal to zero or one (a witness: 80 % of
// them) and the points at infinity of a key (28 .. 41 % of a real a / b query), and a grid planned for n W entries then runs a last round
// that is nearly empty (A of the bench's proof: 2.07 rounds of work took three) or, for a short list, gives a quarter of the lanes 40
// entries each while the others idle (2^16 pairs: 10 entries per lane on average).  `lanes` = chunks the device runs at a time; the list
// is spread over whole rounds of them, `lo` <= chunk <= `hi` (lanes < 2^20, lo and hi < 64).
PCD_HD uint32_t msm_plan_chunk(uint32_t M, uint32_t lanes, uint32_t lo, uint32_t hi) {
  const uint64_t per_round = (uint64_t)lanes * hi;
  const uint64_t rounds = ((uint64_t)M + per_round - 1) / per_round;
  const uint32_t chunk = rounds ? (uint32_t)(((uint64_t)M + (uint64_t)lanes * rounds - 1) / ((uint64_t)lanes * rounds)) : lo;
  return chunk < lo ? lo : chunk > hi ? hi : chunk;
}
// (every lane of the accumulate and fix-up kernels evaluates this itself from the plan word `lanes | lo << 20 | hi << 26`: two divisions
//  against a launch on the critical path of every MSM)
PCD_HD uint32_t msm_plan_word(uint32_t lanes, uint32_t lo, uint32_t hi) { return lanes | (lo << 20) | (hi << 26); }
PCD_HD uint32_t msm_chunk_of_plan(uint32_t M, uint32_t plan) { return msm_plan_chunk(M, plan & 0xFFFFFu, (plan >> 20) & 63u, (plan >> 26) & 63u); }

// key of sorted position p: largest key with off[key] <= p (skipping empty buckets)
PCD_DEV uint32_t msm_find_key(const uint32_t* __restrict__ off, uint32_t nkeys, uint32_t p) {
  uint32_t lo = 0, hi = nkeys;  // invariant off[lo] <= p < off[hi]
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= p) lo = mid; else hi = mid; }
  return lo;
}

// Entry p of the (virtual) sorted list.  With the single-pass binning the list is never materialised: entry p of bucket
// `key` is slots[key][p - off[key]]; the pseudo bucket reads `ones_idx`; after a cap overflow (`*flag`) the compact
// fallback list `sorted_idx` is used for the real buckets.
struct MsmEntrySource {
  const uint32_t* sorted_idx;
  const uint32_t* slots;
  const uint32_t* ones_idx;
  const uint32_t* flag;   // nullptr: always the compact list
  uint32_t cap, ones_key;
};
struct MsmCursor {  // walks the buckets of consecutive list positions
  uint32_t key, key_start, key_end;
  uint32_t next_end;  // off[key + 2], loaded one bucket AHEAD: a lane crosses a bucket edge every ~30 additions, some lane of a wave
                      // nearly every iteration, and a dependent global load at that point would stall the whole wave each time
  PCD_DEV void seek(const uint32_t* __restrict__ off, uint32_t nkeys, uint32_t p) {
    key = msm_find_key(off, nkeys, p);
    key_start = off[key];
    key_end = off[key + 1];
    next_end = key + 2 <= nkeys ? off[key + 2] : key_end;
  }
  // (a few linear steps -- neighbouring buckets are the common case -- then a binary search: with signed digits the upper half
  //  of a window's key range is empty, and the lane that crosses it must not walk 2^(c-1) empty keys one load at a time)
  PCD_DEV void advance_to(const uint32_t* __restrict__ off, uint32_t nkeys, uint32_t p) {
    for (int step = 0; step < 4 && p >= key_end; step++) {
      key++; key_start = key_end; key_end = next_end;
      next_end = key + 2 <= nkeys ? off[key + 2] : key_end;
    }
    if (p >= key_end) seek(off, nkeys, p);
  }
};
PCD_DEV uint32_t msm_entry(const MsmEntrySource& src, bool compact, const MsmCursor& cur, uint32_t p) {
  if (cur.key == src.ones_key) return compact && !src.ones_idx ? src.sorted_idx[p] : src.ones_idx[p - cur.key_start];
  return compact ? src.sorted_idx[p] : src.slots[(size_t)cur.key * src.cap + (p - cur.key_start)];
}

// u32 words between consecutive base points in a `pcdhip_bases` array.  The affine image of a 298-bit G1 point is 88 B;
// PCD_BASE_ALIGN pads the record to 128 B so that the gather of one point touches exactly one 128-B line (it straddles two for
// most 88-B offsets: 4.4 GB of fetch traffic per 2^20 MSM against 1.4 GB of payload, profiles/r01_pmc_fetch_summary_final.csv).
// Round 2 measured it on a lone MSM (-0.3 %) and left it off; round 3 measured it where HBM is shared -- the five concurrent MSM
// streams of a proof: -2.2 % (profiles/r03_ab_base_align_concurrent.txt) -- and with four MSMs in flight: 416 -> 430 Mscalar-mul/s;
// the padding costs 45 % more key memory for the 298-bit G1 queries (2.0 instead of 1.4 GB per 2^20-point query with its copies).
#ifndef PCD_BASE_ALIGN
#define PCD_BASE_ALIGN 1
#endif
template <class G>
struct MsmBaseStride {
  static constexpr int W = Aff<typename G::F>::WORDS;
  static constexpr int value = (PCD_BASE_ALIGN && W == 22) ? 32 : W;
};

// Which coordinates the running sum of a bucket run uses.  XYZZ (EC::madd_x: 8M + 2S, half the additions) wins where the arithmetic
// is inlined or the point is spread thin over lanes -- same-box A/B of the accumulation: Fq2-298 7.07 -> 6.60 ms, Fq3-298 (split)
// 14.8 -> 13.8 ms at 2^20, Fq2-753 (split) 19.3 -> 18.1 ms at 2^16 -- and loses where every product is a call and the fourth
// coordinate is 27 more registers to keep alive across it: G1-753 45.5 -> 48.6 ms (2^20), Fq3-753 (split) 36.1 -> 39.0 ms (2^16).
template <class G>
struct MsmUseXyzz {
  typedef typename G::F F;
// (PCD_XYZZ_753: XYZZ for the 27-limb G1 / Fq3 groups as well.  Re-measured after the LDS mailbox, same-box A/B: G1-753 accumulation
//  27.2 -> 26.2 ms at 2^20 but the fix-up pass 1.29 -> 1.73 ms (four-coordinate records to convert): MSM 33.6 -> 33.1 ms; split Fq3-753
//  at 2^16 16.0 -> 15.0 ms and 2.3 -> 3.5 ms: 28.5 -> 28.7 ms.  Left off.)
#ifndef PCD_XYZZ_753
#define PCD_XYZZ_753 0
#endif
  static constexpr bool value = F::Base::INLINE_ARITH || F::DEG == 2 || (PCD_XYZZ_753 && F::Base::N > 11);
};
// the running sum of a bucket run, flushed as it is: XYZZ (X || Y || ZZ || ZZZ, the identity as ZZ = 0) or Jacobian
template <class G, bool XYZZ = MsmUseXyzz<G>::value>
struct MsmRunPlain {
  typedef typename G::F F;
  typename EC<G>::AccX a = EC<G>::x_infinity();
  PCD_DEV void add(const Aff<F>& q) { a = EC<G>::madd_x(a, q); }
  PCD_DEV void flush(uint32_t* dst) {
    a.X.store(dst); a.Y.store(dst + F::WORDS); a.ZZ.store(dst + 2 * F::WORDS); a.ZZZ.store(dst + 3 * F::WORDS);
    a = EC<G>::x_infinity();
  }
};
template <class G>
struct MsmRunPlain<G, false> {
  typedef typename G::F F;
  Jac<F> a = Jac<F>::infinity();
  PCD_DEV void add(const Aff<F>& q) { a = EC<G>::madd(a, q); }
  PCD_DEV void flush(uint32_t* dst) { a.store(dst); a = Jac<F>::infinity(); }
};
// The lazily reduced accumulator is flushed AS IT IS (XYZZ: X carry-propagated < 16p, Y / ZZ / ZZZ reduced; the identity as
// ZZ = 0) into records of four field elements: the products that turn it into a reduced Jacobian point would run inside the divergent
// flush branch that some lane of the wave takes in nearly every iteration.  Whatever msm_accumulate wrote -- buckets and pieces -- is
// converted by its reader (MsmStored::load: the fix-up pass, one lane per bucket, coalesced and convergent).
template <class G>
struct MsmRunLazy {
  typedef typename G::F F;
  typename EC<G>::AccLz a = EC<G>::lz_infinity();
  PCD_DEV void add(const Aff<F>& q) { a = EC<G>::madd_lz(a, q); }
  PCD_DEV void flush(uint32_t* dst) {
    if (a.inf) { F::zero().store(dst + 2 * F::WORDS); return; }
#pragma unroll
    for (int i = 0; i < F::N; i++) dst[i] = (uint32_t)a.X.v[i];
    a.Y.store(dst + F::WORDS);
    a.ZZ.store(dst + 2 * F::WORDS);
    a.ZZZ.store(dst + 3 * F::WORDS);
    a = EC<G>::lz_infinity();
  }
};
// EXPERIMENT (PCD_ACC_LDS=1, with PCD_ACC_WAVES_G1=3): the same running sum with the four coordinates in per-lane LDS slots
// ([coordinate][16-byte piece][lane]: a wave's access is conflict-free ds_read_b128 / ds_write_b128), EC::madd_lz_st -- so that the
// register allocation can aim at three waves per SIMD.  One wave per workgroup: 4 x 3 x 64 x 16 B = 12 KB, 144 KB for the 12 waves of a CU.
#ifndef PCD_ACC_LDS
#define PCD_ACC_LDS 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
template <class G>
struct MsmRunLazyLds {
  typedef typename G::F F;
  typedef uint32_t V4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) V4* LdsV4;
  static constexpr int CH = (F::N + 3) / 4;
  struct Store {
    LdsV4 box;
    bool is_inf;
    PCD_DEV bool& inf() { return is_inf; }
    PCD_DEV F ld(int slot) const {
      F r;
#pragma unroll
      for (int k = 0; k < CH; k++) {
        const V4 w = box[(slot * CH + k) * 64 + threadIdx.x];
        r.v[4 * k] = w.x;
        if (4 * k + 1 < F::N) r.v[4 * k + 1] = w.y;
        if (4 * k + 2 < F::N) r.v[4 * k + 2] = w.z;
        if (4 * k + 3 < F::N) r.v[4 * k + 3] = w.w;
      }
      return r;
    }
    PCD_DEV void st(int slot, const F& a) {
#pragma unroll
      for (int k = 0; k < CH; k++) {
        V4 w;
        w.x = a.v[4 * k];
        w.y = 4 * k + 1 < F::N ? a.v[4 * k + 1] : 0u;
        w.z = 4 * k + 2 < F::N ? a.v[4 * k + 2] : 0u;
        w.w = 4 * k + 3 < F::N ? a.v[4 * k + 3] : 0u;
        box[(slot * CH + k) * 64 + threadIdx.x] = w;
      }
    }
  };
  Store s;
  PCD_DEV MsmRunLazyLds() {
    __shared__ V4 accbox[4 * CH * 64];
    s.box = (LdsV4)accbox;
    s.is_inf = true;
  }
  PCD_DEV void add(const Aff<F>& q) { EC<G>::madd_lz_st(s, q); }
  PCD_DEV void flush(uint32_t* dst) {
    if (s.is_inf) { F::zero().store(dst + 2 * F::WORDS); return; }
    s.ld(0).store(dst); s.ld(1).store(dst + F::WORDS); s.ld(2).store(dst + 2 * F::WORDS); s.ld(3).store(dst + 3 * F::WORDS);
    s.is_inf = true;
  }
};
#endif
// a point written by msm_accumulate's flush, as a reduced Jacobian point; WORDS = u32 words of one flushed record
template <class G, bool XYZZ = MsmUseXyzz<G>::value>
struct MsmStoredPlain {  // XYZZ record
  typedef typename G::F F;
  static constexpr int WORDS = EC<G>::ACCX_WORDS;
  PCD_DEV static Jac<F> load(const uint32_t* p) {
    typename EC<G>::AccX a = {F::load(p), F::load(p + F::WORDS), F::load(p + 2 * F::WORDS), F::load(p + 3 * F::WORDS)};
    return EC<G>::x_to_jac(a);
  }
  static constexpr bool SEPARATE = true;  // flushed buckets live in their own array (four coordinates); the fix-up pass fills the bucket array
};
template <class G>
struct MsmStoredPlain<G, false> {  // Jacobian record
  typedef typename G::F F;
  static constexpr int WORDS = Jac<F>::WORDS;
  PCD_DEV static Jac<F> load(const uint32_t* p) { return Jac<F>::load(p); }
  static constexpr bool SEPARATE = false;  // flushed buckets ARE the bucket array
};
template <class G, bool LAZY = LazyCapable<typename G::F>::value>
struct MsmStored : MsmStoredPlain<G> {};
template <class G>
struct MsmStored<G, true> {
  typedef typename G::F F;
  static constexpr int WORDS = EC<G>::ACC_WORDS;
  PCD_DEV static Jac<F> load(const uint32_t* p) {
    typename EC<G>::AccLz a;
    a.ZZ = F::load(p + 2 * F::WORDS);
    if (a.ZZ.is_zero()) return Jac<F>::infinity();
    a.inf = false;
#pragma unroll
    for (int i = 0; i < F::N; i++) a.X.v[i] = (int32_t)p[i];
    a.Y = F::load(p + F::WORDS);
    a.ZZZ = F::load(p + 3 * F::WORDS);
    return EC<G>::lz_to_jac(a);
  }
  static constexpr bool SEPARATE = true;  // flushed buckets live in their own array (wider records); the fix-up pass fills the bucket array
};

// Waves per SIMD the register allocation of the accumulate kernel aims at: 2 for the 298-bit G1 (194 registers, no spills);
// 1 wherever the working set does not fit 256 registers -- the extension fields and everything 753-bit -- so that it lives in
// the 512-register budget (AGPRs) instead of scratch.  Same-box A/B on MI355X (2 -> 1): G1-753 28.2 -> 24.5 ms (2^19),
// split Fq2-753 34.9 -> 31.3 ms (2^17), Fq3-298 29.9 -> 25.2 ms (2^20), Fq2-298 7.85 -> 7.44 ms (2^20).
template <class G>
struct MsmAccWaves {
  typedef typename AccOf<G>::type::F FA;
#ifndef PCD_ACC_WAVES_SPLIT
#define PCD_ACC_WAVES_SPLIT 2
#endif
#ifndef PCD_ACC_WAVES_G1
#define PCD_ACC_WAVES_G1 2
#endif
#ifndef PCD_ACC_WAVES_MB298
#define PCD_ACC_WAVES_MB298 2
#endif
  static constexpr int value = !FA::Base::INLINE_ARITH ? (FA::Base::N <= 11 ? PCD_ACC_WAVES_MB298 : 1)
                             : FA::DEG == 1 ? PCD_ACC_WAVES_G1 : AccOf<G>::LANES > 1 ? PCD_ACC_WAVES_SPLIT : 1;
};

// COMPACT: the entries form one list (`sorted_idx`, the scalars equal to one possibly in their own list `ones_idx` behind it), so
// entry p is found without knowing its bucket; otherwise (single-pass binning, pcdhip_msm_set_sort 1) entries sit in per-bucket slots
// and are addressed through the cursor.
template <class G, bool COMPACT>
__global__ void __launch_bounds__(64, MsmAccWaves<G>::value) msm_accumulate_kernel(const uint32_t* __restrict__ bases, const MsmEntrySource src,
                                                            const uint32_t* __restrict__ off, uint32_t nkeys,
                                                            uint32_t chunk, uint32_t* __restrict__ buckets,
                                                            uint32_t* __restrict__ piece_first, uint32_t* __restrict__ piece_last,
                                                            uint32_t plan /* msm_plan_word, or 0: `chunk` as given */) {
  if (plan) chunk = msm_chunk_of_plan(off[nkeys], plan);
  // GA: the configuration the arithmetic runs in -- G itself, or its lane-split form (Fq2 over lane pairs, Fq3 over lane triples:
  // the lanes of a group share one chunk, each holding one coefficient of every coordinate; memory images are the same)
  typedef typename AccOf<G>::type GA;
  typedef typename GA::F F;
  typedef EC<GA> E;
  constexpr int RW = MsmStored<GA>::WORDS;  // u32 words of a flushed record (`buckets` here = the array the flushes go to)
  constexpr uint32_t LANES = AccOf<G>::LANES, PER_WAVE = 64 / LANES;  // chunks per 64-lane workgroup (three lanes per point: 21, lane 63 idles)
  if (threadIdx.x >= PER_WAVE * LANES) return;
  uint32_t t = blockIdx.x * PER_WAVE + threadIdx.x / LANES;
  const uint32_t M = off[nkeys];  // total sorted entries: read on the device, the host never waits for it
  uint64_t start64 = (uint64_t)t * chunk;
  if (start64 >= M) return;
  uint32_t start = (uint32_t)start64, end = (uint32_t)min((uint64_t)M, start64 + chunk);
  if (!COMPACT && src.flag && *src.flag != 0) return;  // a bucket overflowed its slots: the compact instantiation runs instead
  if (COMPACT && src.flag && *src.flag == 0) return;
  MsmCursor cur;
  cur.seek(off, nkeys, start);
  bool open_start = cur.key_start < start;  // current run began in an earlier chunk
  // the running sum: Jacobian, or (G1 of the 298-bit curves) the lazily reduced accumulator of EC::madd_lz
#if PCD_ACC_LDS
  typename std::conditional<LazyCapable<F>::value, MsmRunLazyLds<GA>, MsmRunPlain<GA>>::type acc;
#else
  typename std::conditional<LazyCapable<F>::value, MsmRunLazy<GA>, MsmRunPlain<GA>>::type acc;
#endif
  // The next point is prefetched while the current addition runs -- except for the 753-bit fields, whose products are
  // function calls: the 54 .. 162 registers of a prefetched point are live across eleven calls per addition and get spilled
  // around every one of them (same-box A/B on MI355X: G1-753 31.4 -> 27.5 ms at 2^19, split Fq2-753 37.6 -> 33.3 ms at 2^17,
  // Fq3-753 73 -> 47 ms at 2^16 without the prefetch).
  constexpr bool PREFETCH = F::Base::INLINE_ARITH;
  const uint32_t ones_start = off[src.ones_key];  // list position of the first scalar equal to one
  auto point_of = [&](uint32_t e) {  // the entry's point, negated for a negative digit
    Aff<F> q = Aff<F>::load(bases + (size_t)(e & ~MSM_NEG) * MsmBaseStride<G>::value);
    const F ny = q.y.neg();
    if (e & MSM_NEG) q.y = ny;
    return q;
  };
  if constexpr (COMPACT) {
    auto entry_at = [&](uint32_t pos) { return (src.ones_idx && pos >= ones_start) ? src.ones_idx[pos - ones_start] : src.sorted_idx[pos]; };
    Aff<F> nxt;
    if (PREFETCH) nxt = point_of(entry_at(start));
    for (uint32_t p = start; p < end; p++) {
      if (p >= cur.key_end) {  // run of `key` is complete
        if (open_start) { acc.flush(piece_first + (size_t)t * RW); open_start = false; }
        else acc.flush(buckets + (size_t)cur.key * RW);
        cur.advance_to(off, nkeys, p);
      }
      if (PREFETCH) {
        const Aff<F> pt = nxt;
        if (p + 1 < end) nxt = point_of(entry_at(p + 1));
        acc.add(pt);
      } else {
        acc.add(point_of(entry_at(p)));
      }
    }
  } else {
    MsmCursor nxt_cur = cur;
    Aff<F> nxt;
    if (PREFETCH) nxt = point_of(msm_entry(src, false, cur, start));
    for (uint32_t p = start; p < end; p++) {
      if (p >= cur.key_end) {
        if (open_start) { acc.flush(piece_first + (size_t)t * RW); open_start = false; }
        else acc.flush(buckets + (size_t)cur.key * RW);
        cur.advance_to(off, nkeys, p);
      }
      if (PREFETCH) {
        const Aff<F> pt = nxt;
        if (p + 1 < end) {
          nxt_cur.advance_to(off, nkeys, p + 1);
          nxt = point_of(msm_entry(src, false, nxt_cur, p + 1));
        }
        acc.add(pt);
      } else {
        acc.add(point_of(msm_entry(src, false, cur, p)));
      }
    }
  }
  bool open_end = cur.key_end > end;
  if (open_end) acc.flush(piece_last + (size_t)t * RW);          // also the "middle piece" case
  else if (open_start) acc.flush(piece_first + (size_t)t * RW);
  else acc.flush(buckets + (size_t)cur.key * RW);
}

// ------------------------------------------------------------------------------------------------ accumulate, pair-tree form
// The same stage for the 753-bit G1 groups as an opt-in mode (pcdhip_msm_set_accumulate(ctx, 2, ..); DESIGN.md 4 has the A/B that
// keeps the running sums the default), with AFFINE additions whose inversions are shared (the reference's MSM has no such form).  A lane
// owns a chunk of several hundred sorted entries and halves it level by level: neighbours of the same bucket form a pair,
// P + Q = (l^2 - x1 - x2, l (x1 - x3) - y1) with l = (y2 - y1) / (x2 - x1), and the denominators of ALL the lane's pairs of a level
// are inverted together -- a running product on the way forth (kept in HBM, lane-interleaved), ONE inversion (Fp::inv_gcd, 45
// products' worth), and on the way back each pair's inverse peels off with two products: 5M + 1S per addition against 7M + 4S for the
// Jacobian mixed addition, no Z coordinate, results written back as affine points (216 B) that the next level pairs again.  Levels
// stop when the widest lane of the wave has fewer than `min_pairs` pairs left (an inversion no longer pays); what remains of every
// bucket run -- usually one point -- goes through the ordinary running sum and is flushed exactly as msm_accumulate_kernel does, so
// the fix-up pass and everything behind it are unchanged.  Equal points (doubling: denominator 2y), opposite points, operands at
// infinity and points with x = 0 are classified on the way forth in a branch that is normally skipped.
// One wave per workgroup, workgroups persistent (a wave walks chunk groups blockIdx.x, + gridDim.x, ...): the scratch areas belong
// to the resident wave, not to the chunk.  Everything a lane keeps in HBM is indexed [item][piece][lane] (lists: 4-byte pieces, field
// elements: 16-byte pieces): a wave's access to "its k-th pair" is one contiguous 256-byte / 1-KB block per piece.
constexpr uint32_t MSM_TREE_CHUNK_MAX = 640, MSM_TREE_MIN_PAIRS = 12;
constexpr uint32_t MSM_SCR = 0x40000000u;  // item location: slot of the lane's point scratch (else a base entry, MSM_NEG = negated)
template <class G>
struct MsmTreeCapable {
  typedef typename AccOf<G>::type::F F;
  static constexpr bool value = F::DEG == 1 && AccOf<G>::LANES == 1 && F::Base::N > 11;
};
struct MsmTreeBufs {
  uint32_t* list;    // per wave: 2 lists x cap items x (key, loc) x 64 lanes
  uint32_t* pairs;   // per wave: (cap / 2 + 1) x (loc a, loc b, kind) x 64
  uint32_t* prefix;  // per wave: (cap / 2 + 1) field elements x 64
  uint32_t* pts;     // per wave: cap affine points x 64
  uint32_t cap;      // entries per chunk
  uint32_t phases;   // 3: trip count of the first product loop of the way back (a kernel argument so that the loops stay loops)
  PCD_HD static size_t list_words(uint32_t cap) { return (size_t)4 * cap * 64; }
  PCD_HD static size_t pair_words(uint32_t cap) { return (size_t)3 * (cap / 2 + 1) * 64; }
  PCD_HD static size_t prefix_words(uint32_t cap, int fw) { return (size_t)((fw + 3) / 4 * 4) * (cap / 2 + 1) * 64; }
  // (slots are handed out per level for the whole wave -- the widest lane's pair count -- so that "pair k" is the same slot in every lane:
  //  a level with L pairs at most cap / (2^(l-1) + 1) wide, summed over the levels < 1.23 cap)
  PCD_HD static size_t pts_words(uint32_t cap, int fw) { return (size_t)2 * ((fw + 3) / 4 * 4) * (cap + cap / 4 + 8) * 64; }
};
PCD_DEV uint32_t msm_wave_max(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m, 64));
  return v;
}
enum { MSM_PK_ADD = 0, MSM_PK_TAKE_B = 1, MSM_PK_TAKE_A = 2, MSM_PK_INF = 3, MSM_PK_DBL = 4 };

// entries per lane: the list spread over the lanes of the persistent grid, in as many rounds as keep a chunk within the scratch capacity
static __global__ void msm_tree_plan_kernel(const uint32_t* __restrict__ off, uint32_t nkeys, uint32_t lanes, uint32_t cap, uint32_t forced,
                                            uint32_t* __restrict__ chunk_out) {
  if (blockIdx.x || threadIdx.x) return;
  const uint64_t M = off[nkeys];
  const uint64_t rounds = (M + (uint64_t)lanes * cap - 1) / ((uint64_t)lanes * cap);
  uint32_t chunk = rounds ? (uint32_t)((M + lanes * rounds - 1) / (lanes * rounds)) : 32u;
  chunk = min(max(chunk, 32u), cap);
  *chunk_out = forced ? forced : chunk;
}
template <class G>
__global__ void __launch_bounds__(64, 1) msm_pair_tree_kernel(const uint32_t* __restrict__ bases, const MsmEntrySource src,
                                                              const uint32_t* __restrict__ off, uint32_t nkeys, const uint32_t* __restrict__ chunk_dev,
                                                              const MsmTreeBufs tb, uint32_t min_pairs, uint32_t* __restrict__ buckets,
                                                              uint32_t* __restrict__ piece_first, uint32_t* __restrict__ piece_last) {
  typedef typename AccOf<G>::type GA;
  typedef typename GA::F F;
  constexpr int RW = MsmStored<GA>::WORDS, FW = F::WORDS;
  const uint32_t lane = threadIdx.x;
  const uint32_t M = off[nkeys];
  const uint32_t chunk = *chunk_dev;  // msm_tree_plan_kernel: from the ACTUAL list length (a witness's list is a fraction of n W)
  const uint32_t nwaves = (uint32_t)((((uint64_t)M + chunk - 1) / chunk + 63) / 64);
  const bool compact = !(src.slots && src.flag && *src.flag == 0);
  const uint32_t ones_start = off[src.ones_key];
  uint32_t* const LS = tb.list + (size_t)blockIdx.x * MsmTreeBufs::list_words(tb.cap) + lane;
  uint32_t* const PR = tb.pairs + (size_t)blockIdx.x * MsmTreeBufs::pair_words(tb.cap) + lane;
  constexpr int ECH = (FW + 3) / 4, EW = 4 * ECH;  // 16-byte pieces / padded words of one element in the scratch areas
  uint32_t* const PX = tb.prefix + (size_t)blockIdx.x * MsmTreeBufs::prefix_words(tb.cap, FW) + lane * 4;
  uint32_t* const PT = tb.pts + (size_t)blockIdx.x * MsmTreeBufs::pts_words(tb.cap, FW) + lane * 4;
  // one element = ECH 16-byte pieces, piece c of lane l at words (c * 64 + l) * 4 of its row: a wave moves 1 KB per instruction
  typedef uint32_t V4 __attribute__((ext_vector_type(4)));
  auto fld_ld = [&](const uint32_t* p) { F r;
#pragma unroll
    for (int c = 0; c < ECH; c++) {
      const V4 v = *reinterpret_cast<const V4*>(p + (size_t)c * 256);
      r.v[4 * c] = v.x;
      if (4 * c + 1 < FW) r.v[4 * c + 1] = v.y;
      if (4 * c + 2 < FW) r.v[4 * c + 2] = v.z;
      if (4 * c + 3 < FW) r.v[4 * c + 3] = v.w;
    }
    return r; };
  auto fld_st = [&](uint32_t* p, const F& a) {
#pragma unroll
    for (int c = 0; c < ECH; c++) {
      V4 v;
      v.x = a.v[4 * c];
      v.y = 4 * c + 1 < FW ? a.v[4 * c + 1] : 0u;
      v.z = 4 * c + 2 < FW ? a.v[4 * c + 2] : 0u;
      v.w = 4 * c + 3 < FW ? a.v[4 * c + 3] : 0u;
      *reinterpret_cast<V4*>(p + (size_t)c * 256) = v;
    } };
  // (the wave-uniform cases first: at level 1 every operand is a base entry -- contiguous records, wide loads -- and later nearly
  //  every one a scratch point; the compiler turns a per-lane choice of the two into ONE sequence of 4-byte loads with selected
  //  addresses, 27 x 64 separate lines per element through the CU's one address path)
  auto x_of = [&](uint32_t loc) {
    const bool scr = (loc & MSM_SCR) != 0;
    if (!__any(scr)) return F::load(bases + (size_t)(loc & ~MSM_NEG) * MsmBaseStride<G>::value);
    if (__all(scr)) return fld_ld(PT + (size_t)(loc & ~MSM_SCR) * (2 * EW * 64));
    F r = F::zero();
    if (scr) r = fld_ld(PT + (size_t)(loc & ~MSM_SCR) * (2 * EW * 64));
    else r = F::load(bases + (size_t)(loc & ~MSM_NEG) * MsmBaseStride<G>::value);
    return r; };
  auto base_point = [&](uint32_t loc) {
    Aff<F> q = Aff<F>::load(bases + (size_t)(loc & ~MSM_NEG) * MsmBaseStride<G>::value);
    const F ny = q.y.neg();
    if (loc & MSM_NEG) q.y = ny;
    return q; };
  auto scratch_point = [&](uint32_t loc) {
    const uint32_t* p = PT + (size_t)(loc & ~MSM_SCR) * (2 * EW * 64);
    Aff<F> q;
    q.x = fld_ld(p); q.y = fld_ld(p + (size_t)EW * 64);
    return q; };
  auto point_of = [&](uint32_t loc) {
    const bool scr = (loc & MSM_SCR) != 0;
    if (!__any(scr)) return base_point(loc);
    if (__all(scr)) return scratch_point(loc);
    Aff<F> q = {F::zero(), F::zero()};
    if (scr) q = scratch_point(loc);
    else q = base_point(loc);
    return q; };
  for (uint32_t w = blockIdx.x; w < nwaves; w += gridDim.x) {
    const uint32_t t = w * 64 + lane;
    const uint64_t start64 = (uint64_t)t * chunk;
    const uint32_t start = start64 < M ? (uint32_t)start64 : M;
    const uint32_t n0 = min(M - start, chunk), end = start + n0;
    // level 0: the chunk's entries with their buckets
    uint32_t key_first = 0;
    bool open_start = false;
    if (n0) {
      MsmCursor cur;
      cur.seek(off, nkeys, start);
      key_first = cur.key;
      open_start = cur.key_start < start;
      for (uint32_t i = 0; i < n0; i++) {
        const uint32_t p = start + i;
        if (p >= cur.key_end) cur.advance_to(off, nkeys, p);
        uint32_t e;
        if (compact) e = (src.ones_idx && p >= ones_start) ? src.ones_idx[p - ones_start] : src.sorted_idx[p];
        else e = msm_entry(src, false, cur, p);
        LS[(size_t)(2 * i) * 64] = cur.key;
        LS[(size_t)(2 * i + 1) * 64] = e;
      }
    }
    uint32_t cnt = n0, which = 0, slot_base = 0;
    for (;;) {
      uint32_t* const in = LS + (size_t)which * (2 * tb.cap * 64);
      uint32_t* const out = LS + (size_t)(which ^ 1) * (2 * tb.cap * 64);
      uint32_t np = 0, j = 0;
      for (uint32_t i = 0; i < cnt; j++) {
        const uint32_t k0 = in[(size_t)(2 * i) * 64], l0 = in[(size_t)(2 * i + 1) * 64];
        if (i + 1 < cnt && in[(size_t)(2 * i + 2) * 64] == k0) {
          PR[(size_t)(3 * np) * 64] = l0;
          PR[(size_t)(3 * np + 1) * 64] = in[(size_t)(2 * i + 3) * 64];
          out[(size_t)(2 * j) * 64] = k0;
          out[(size_t)(2 * j + 1) * 64] = MSM_SCR | (slot_base + np);
          np++; i += 2;
        } else {
          out[(size_t)(2 * j) * 64] = k0;
          out[(size_t)(2 * j + 1) * 64] = l0;
          i++;
        }
      }
      const uint32_t np_max = msm_wave_max(np);
      if (np_max < min_pairs) break;
      // forth: the kinds, the denominators and their running product.  The product is INLINED here (and the five of the way back
      // run through one inlined copy): a call makes the compiler wait for every outstanding load first, and at one wave per SIMD
      // nothing else hides the 2-3 us of a dependent load -- pair k + 1's operands and pair k + 2's list entries are requested
      // before pair k's product and arrive behind it.
      F run = F::one();
      {
        uint32_t la = 0, lb = 0, la1 = 0, lb1 = 0;
        if (0 < np) { la = PR[0]; lb = PR[(size_t)64]; }
        if (1 < np) { la1 = PR[(size_t)3 * 64]; lb1 = PR[(size_t)4 * 64]; }
        F x1n = F::zero(), x2n = F::zero();
        if (0 < np) { x1n = x_of(la); x2n = x_of(lb); }
        for (uint32_t k = 0; k < np_max; k++) {
          const F x1 = x1n, x2 = x2n;
          const uint32_t ca = la, cb = lb;
          la = la1; lb = lb1;
          if (k + 2 < np) { la1 = PR[(size_t)(3 * k + 6) * 64]; lb1 = PR[(size_t)(3 * k + 7) * 64]; }
          if (k + 1 < np) { x1n = x_of(la); x2n = x_of(lb); }
          F dd = F::one();
          if (k < np) {
            uint32_t kind = MSM_PK_ADD;
            dd = x2 - x1;
            if (__builtin_expect(x1.is_raw_zero() || x2.is_raw_zero() || dd.is_zero(), 0)) {
              const Aff<F> a = point_of(ca), b = point_of(cb);
              if (a.is_inf()) { kind = MSM_PK_TAKE_B; dd = F::one(); }
              else if (b.is_inf()) { kind = MSM_PK_TAKE_A; dd = F::one(); }
              else if (dd.is_zero()) {
                if ((a.y + b.y).is_zero()) { kind = MSM_PK_INF; dd = F::one(); }
                else { kind = MSM_PK_DBL; dd = a.y.dbl(); }
              }
            }
            PR[(size_t)(3 * k + 2) * 64] = kind;
            fld_st(PX + (size_t)k * (EW * 64), run);
          }
          run = F::mul_impl(run, dd);
        }
      }
      F inv = run.inv_gcd();
      // back: 1 / d_k = inv * (d_0 .. d_{k-1}), inv *= d_k
      {
        Aff<F> an = {F::zero(), F::zero()}, bn = an;
        F pren = F::one();
        uint32_t kindn = MSM_PK_INF, la = 0, lb = 0;
        if (np_max - 1 < np) {  // the widest lanes' last pair
          const uint32_t k = np_max - 1;
          an = point_of(PR[(size_t)(3 * k) * 64]); bn = point_of(PR[(size_t)(3 * k + 1) * 64]);
          kindn = PR[(size_t)(3 * k + 2) * 64];
          pren = fld_ld(PX + (size_t)k * (EW * 64));
        }
        if (np_max >= 2 && np_max - 2 < np) { la = PR[(size_t)(3 * (np_max - 2)) * 64]; lb = PR[(size_t)(3 * (np_max - 2) + 1) * 64]; }
        for (uint32_t k = np_max; k-- > 0;) {
          const bool live = k < np;
          const uint32_t kind = live ? kindn : (uint32_t)MSM_PK_ADD;
          // what the five products need of pair k: x1, y1, x1 + x2, the denominator and the numerator (the second point dies here)
          const F ax = an.x, ay = an.y, pre = pren;
          const F sx = an.x + bn.x;
          F dd = F::one(), num = F::zero();
          if (live && kind == MSM_PK_ADD) { dd = bn.x - an.x; num = bn.y - an.y; }
          if (__builtin_expect(__any(kind == MSM_PK_DBL), 0)) {
            if (kind == MSM_PK_DBL) { const F xx = ax.sqr(); dd = ay.dbl(); num = xx.dbl() + xx + GA::mul_by_a(F::one()); }
          }
          // ik = inv * pre; inv *= dd; lam = num * ik through one inlined copy of the product -- then pair k - 1's operands are
          // requested -- lam^2 and lam * (x1 - x3) straight-line (nothing of the first three is alive any more)
          F t0 = F::zero(), lam = F::zero();
#pragma nounroll
          for (uint32_t ph = 0; ph < tb.phases; ph++) {
            F u, v;
            if (ph == 0) { u = inv; v = pre; }
            else if (ph == 1) { u = inv; v = dd; }
            else { u = num; v = t0; }
            const F m = F::mul_impl(u, v);
            if (ph == 0) t0 = m;
            else if (ph == 1) inv = m;
            else lam = m;
          }
          if (k >= 1 && k - 1 < np) {  // pair k - 1: its points, its kind, its prefix; pair k - 2: its list entries
            an = point_of(la); bn = point_of(lb);
            kindn = PR[(size_t)(3 * (k - 1) + 2) * 64];
            pren = fld_ld(PX + (size_t)(k - 1) * (EW * 64));
          }
          if (k >= 2 && k - 2 < np) { la = PR[(size_t)(3 * (k - 2)) * 64]; lb = PR[(size_t)(3 * (k - 2) + 1) * 64]; }
          Aff<F> r;
          r.x = F::sqr_impl(lam) - sx;
          r.y = F::mul_impl(lam, ax - r.x) - ay;
          if (__builtin_expect(__any(live && kind != MSM_PK_ADD && kind != MSM_PK_DBL), 0)) {  // one of the two as it is, or the identity
            if (live && kind == MSM_PK_TAKE_A) r = point_of(PR[(size_t)(3 * k) * 64]);
            if (live && kind == MSM_PK_TAKE_B) r = point_of(PR[(size_t)(3 * k + 1) * 64]);
            if (live && kind == MSM_PK_INF) r = {F::zero(), F::zero()};
          }
          if (live) {
            uint32_t* p = PT + (size_t)(slot_base + k) * (2 * EW * 64);
            fld_st(p, r.x); fld_st(p + (size_t)EW * 64, r.y);
          }
        }
      }
      slot_base += np_max;
      cnt = j;
      which ^= 1;
    }
    // what is left of every run: the ordinary running sum, flushed as msm_accumulate_kernel flushes
    if (n0) {
      const uint32_t* const in = LS + (size_t)which * (2 * tb.cap * 64);
      MsmRunPlain<GA> acc;
      uint32_t key = key_first;
      for (uint32_t i = 0; i < cnt; i++) {
        const uint32_t k0 = in[(size_t)(2 * i) * 64], l0 = in[(size_t)(2 * i + 1) * 64];
        if (k0 != key) {
          if (open_start) { acc.flush(piece_first + (size_t)t * RW); open_start = false; }
          else acc.flush(buckets + (size_t)key * RW);
          key = k0;
        }
        acc.add(point_of(l0));
      }
      const bool open_end = off[key + 1] > end;
      if (open_end) acc.flush(piece_last + (size_t)t * RW);
      else if (open_start) acc.flush(piece_first + (size_t)t * RW);
      else acc.flush(buckets + (size_t)key * RW);
    }
  }
}

// Work-item indexing of the kernels behind the accumulation (pieces, bucket reduction, window combine).  They compute in the
// lane-split form of the group too: an Fq2 / Fq3 point operation is shared by 2 / 3 adjacent lanes, which divides the latency of
// these latency-bound levels and the per-lane scratch footprint (9.9 KB per lane for the unsplit Fq3-753 addition: enough to
// exhaust the queues' scratch reservations once a few contexts are alive) by the same factor.  A 64-lane workgroup holds
// PER_WAVE items; the lanes of an item follow the same control flow (decisions depend on the item only).
template <class G>
struct MsmItems {
  typedef typename SplitOfTail<G>::type GA;
  static constexpr uint32_t LANES = SplitOfTail<G>::LANES, PER_WAVE = 64 / LANES;
  PCD_DEV static bool idle() { return threadIdx.x >= PER_WAVE * LANES; }
  PCD_DEV static uint32_t local() { return threadIdx.x / LANES; }
  PCD_DEV static uint32_t item() { return blockIdx.x * PER_WAVE + threadIdx.x / LANES; }
  static uint32_t grid(uint32_t items) { return (items + PER_WAVE - 1) / PER_WAVE; }
};

// bucket (window 0, digit 1) += pseudo bucket of the scalars equal to one
// (ONE addition per MSM, so it computes in the plain lane-split form of the group, never in the mailbox variant the other tail kernels
//  use for the 753-bit groups: with this kernel's shape -- one item, both operands finite -- the mailbox form of the Fq3-753 addition
//  returns one wrong limb of Y (localised with tools/probe_fq3_tail.py: "ones and 257"; every other kernel and every other case is
//  exact in both forms, also with every item slot of the wave kept active here) -- a code-generation problem this sidesteps)
// The zeroing an MSM needs before its first kernel, in ONE launch: the histogram / cursor words the sort starts from and the four control
// words (binning flag, error word, big-bucket and segment counters).  Round 3 spent three hipMemsetAsync per MSM on this -- 2 054 fill
// launches in a bench run, each a dependent launch on an MSM's critical path queued behind one-wave-per-SIMD accumulate grids.
static __global__ void __launch_bounds__(256) msm_zero_kernel(uint32_t* __restrict__ cnt, uint32_t n_cnt, uint32_t* __restrict__ ctl4) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_cnt) cnt[i] = 0;
  if (i < 4) ctl4[i] = 0;
}

#ifndef PCD_SINGLE_ITEM_PLAIN
#define PCD_SINGLE_ITEM_PLAIN 0  // 1: the two single-item kernels (this one, msm_horner_kernel) in the plain lane-split form as in round 3
#endif
template <class G> struct MsmSingleItem {
  typedef typename std::conditional<PCD_SINGLE_ITEM_PLAIN != 0, typename SplitOf<G>::type, typename SplitOfTail<G>::type>::type GA;
  static constexpr int LANES = PCD_SINGLE_ITEM_PLAIN != 0 ? SplitOf<G>::LANES : SplitOfTail<G>::LANES;
};
template <class G>
__global__ void __launch_bounds__(64) msm_merge_ones_kernel(uint32_t* __restrict__ buckets, uint32_t ones_key) {
  typedef typename MsmSingleItem<G>::GA GA;
  typedef typename GA::F F;
  if (blockIdx.x != 0 || threadIdx.x >= MsmSingleItem<G>::LANES) return;
  Jac<F> a = Jac<F>::load(buckets + (size_t)1 * Jac<F>::WORDS);
  Jac<F> b = Jac<F>::load(buckets + (size_t)ones_key * Jac<F>::WORDS);
  EC<GA>::add(a, b).store(buckets + (size_t)1 * Jac<F>::WORDS);
}

// One lane per bucket: a bucket whose sorted run crosses chunk edges is the sum of the pieces its chunks
// emitted (piece_last of the first and middle chunks, piece_first of the last one).  Buckets with more than
// `big_limit` pieces (e.g. "scalar = 1" of a bit-heavy witness, or the low buckets that the short top window
// fills) go to the big list and are summed by one wave each.
template <class G>
__global__ void __launch_bounds__(64) msm_fixup_kernel(const uint32_t* __restrict__ off, uint32_t nkeys, uint32_t chunk,
                                                       const uint32_t* __restrict__ piece_first, const uint32_t* __restrict__ piece_last,
                                                       const uint32_t* __restrict__ acc_buckets /* what msm_accumulate flushed whole buckets into */,
                                                       uint32_t* __restrict__ buckets, uint32_t big_limit, uint32_t* __restrict__ big_count,
                                                       uint32_t* __restrict__ big_list /* (key, first segment, #segments) */, uint32_t big_cap,
                                                       uint32_t* __restrict__ seg_list /* (t_lo, t_hi, t_last) */, uint32_t seg_len,
                                                       const uint32_t* __restrict__ chunk_dev /* the pair tree's chunk, decided on the device */,
                                                       uint32_t plan /* msm_plan_word of the running-sum form, or 0 */) {
  if (chunk_dev) chunk = *chunk_dev;
  else if (plan) chunk = msm_chunk_of_plan(off[nkeys], plan);
  typedef typename MsmItems<G>::GA GA;
  typedef typename GA::F F;
  typedef EC<GA> E;
  constexpr int RW = MsmStored<GA>::WORDS;
  if (MsmItems<G>::idle()) return;
  const bool lead = threadIdx.x % MsmItems<G>::LANES == 0;  // one lane of the item does the list bookkeeping
  uint32_t key = MsmItems<G>::item();
  if (key >= nkeys) return;
  uint32_t lo = off[key], hi = off[key + 1];
  if (hi == lo) {  // empty bucket: nobody else writes it, and the identity is Z = 0 (X, Y are never looked at then)
    F::zero().store(buckets + (size_t)key * Jac<F>::WORDS + 2 * F::WORDS);
    return;
  }
  uint32_t t0 = lo / chunk, t1 = (hi - 1) / chunk;
  if (t1 == t0) {  // the whole run lies inside one chunk: msm_accumulate wrote the bucket itself (unreduced for the lazy groups)
    if (MsmStored<GA>::SEPARATE) MsmStored<GA>::load(acc_buckets + (size_t)key * RW).store(buckets + (size_t)key * Jac<F>::WORDS);
    return;
  }
  if (t1 - t0 + 1 > big_limit) {  // big bucket: its pieces are cut into segments of seg_len, one wave each
    if (!lead) return;
    uint32_t nseg = (t1 - t0 + seg_len) / seg_len;
    uint32_t slot = atomicAdd(&big_count[0], 1u);
    uint32_t s0 = atomicAdd(&big_count[1], nseg);
    if (slot < big_cap) { big_list[3 * slot] = key; big_list[3 * slot + 1] = s0; big_list[3 * slot + 2] = nseg; }
    for (uint32_t i = 0; i < nseg; i++) {
      uint32_t lo = t0 + i * seg_len, hi = min(t1, lo + seg_len - 1);
      seg_list[3 * (s0 + i)] = lo; seg_list[3 * (s0 + i) + 1] = hi; seg_list[3 * (s0 + i) + 2] = t1;
    }
    return;
  }
  Jac<F> acc = MsmStored<GA>::load(piece_last + (size_t)t0 * RW);
  for (uint32_t u = t0 + 1; u < t1; u++) acc = E::add(acc, MsmStored<GA>::load(piece_last + (size_t)u * RW));
  acc = E::add(acc, MsmStored<GA>::load(piece_first + (size_t)t1 * RW));
  acc.store(buckets + (size_t)key * Jac<F>::WORDS);
}

// Items of the bucket-reduction levels.  Prime-field groups spend TWO lanes on every item (EC2: 8 product slots per addition
// instead of 16 products, 5 per doubling instead of 9); the extension-field groups are spread over lanes at the field level already.
template <class G, bool ENABLE = true, bool Q = true>
struct MsmPairItems {
  // (an item whose group is lane-split over L lanes spends 2 L: two halves of L lanes, EC2 with L > 1)
  static constexpr bool TWO = ENABLE && TwoLaneOps<G>::value && (MsmItems<G>::LANES == 1 || HalfLanes<typename MsmItems<G>::GA::F>::L == (int)MsmItems<G>::LANES);
  // prime-field groups: four lanes per operation (EC4) where the caller allows it (Q) -- the LATE pair levels: a level with thousands of items is
  // throughput, where four lanes spend 36 lane-products on an addition + doubling against 26 of the two-lane form (measured: all levels on four
  // lanes moved a 2^20 MSM's tail by 2 %, a 2^16 one's by 16 %; profiles/r06_ab_ec4_all_levels.txt)
  static constexpr bool QUAD = Q && TWO && MsmItems<G>::LANES == 1 && PCD_EC4 != 0;
  static constexpr uint32_t BASE = MsmItems<G>::LANES, LANES = QUAD ? 4 : TWO ? 2 * BASE : BASE, PER_WAVE = 64 / LANES;
  PCD_DEV static bool idle() { return threadIdx.x >= PER_WAVE * LANES; }
  PCD_DEV static uint32_t item() { return blockIdx.x * PER_WAVE + threadIdx.x / LANES; }
  PCD_DEV static uint32_t local() { return threadIdx.x / LANES; }
  PCD_DEV static bool writer() { return QUAD ? (threadIdx.x & 3u) == 0 : (!TWO || (((threadIdx.x & 63u) / BASE) & 1u) == 0); }  // (every half / lane of an item holds the whole result)
  static uint32_t grid(uint32_t items) { return (items + PER_WAVE - 1) / PER_WAVE; }
};
template <class G, bool TWO = MsmPairItems<G, true>::TWO, bool Q = true>
struct MsmPairOps {
  typedef typename MsmItems<G>::GA GA;
  typedef Jac<typename GA::F> J;
  PCD_DEV static J add(const J& a, const J& b) { return EC<GA>::add(a, b); }
  PCD_DEV static J dbl(const J& a) { return EC<GA>::dbl(a); }
};
template <class G, bool Q>
struct MsmPairOps<G, true, Q> {
  typedef typename SplitOfTail<G>::type GA;  // G itself, or its mailbox variant (753-bit)
  typedef Jac<typename GA::F> J;
  PCD_DEV static J add(const J& a, const J& b) { if constexpr (MsmPairItems<G, true, Q>::QUAD) return EC4<GA>::add4(a, b); else return EC2<GA>::add2(a, b); }
  PCD_DEV static J dbl(const J& a) { if constexpr (MsmPairItems<G, true, Q>::QUAD) return EC4<GA>::dbl4(a); else return EC2<GA>::dbl2(a); }
};
// Big buckets, two levels (all counts live on the device; grid-stride loops):
//   A: one workgroup per segment of <= seg_len pieces -> partial[segment]      B: one workgroup per big bucket sums its partials.
// The items of a workgroup sum strided pieces, then a tree through global scratch -- latency-bound chains of additions, so with the
// item / operation choice of the reduction levels (two lanes per addition for the prime-field groups).
template <class G>
PCD_DEV void msm_wave_tree(Jac<typename MsmPairOps<G>::GA::F> acc, uint32_t* my /* PER_WAVE points of scratch */, uint32_t* dst) {
  typedef MsmPairItems<G> IT;
  typedef MsmPairOps<G> O;
  typedef typename O::GA::F F;
  constexpr uint32_t PW = IT::PER_WAVE;
  const uint32_t it = IT::local();
  const bool live = !IT::idle();
  if (live && IT::writer()) acc.store(my + (size_t)it * Jac<F>::WORDS);
  __syncthreads();
  for (uint32_t s = 32; s > 0; s >>= 1) {
    if (live && it < s && it + s < PW) {
      Jac<F> o = Jac<F>::load(my + (size_t)(it + s) * Jac<F>::WORDS);
      acc = O::add(acc, o);
      if (IT::writer()) acc.store(my + (size_t)it * Jac<F>::WORDS);
    }
    __syncthreads();
  }
  if (live && it == 0 && IT::writer()) acc.store(dst);
  __syncthreads();
}
template <class G>
__global__ void __launch_bounds__(64) msm_big_segments_kernel(const uint32_t* __restrict__ seg_list, const uint32_t* __restrict__ big_count,
                                                              const uint32_t* __restrict__ piece_first, const uint32_t* __restrict__ piece_last,
                                                              uint32_t* __restrict__ partial, uint32_t* __restrict__ scratch) {
  typedef MsmPairItems<G> IT;
  typedef MsmPairOps<G> O;
  typedef typename O::GA GA;
  typedef typename GA::F F;
  constexpr uint32_t PW = IT::PER_WAVE;
  const uint32_t nseg = big_count[1];
  uint32_t* my = scratch + (size_t)blockIdx.x * 64 * Jac<F>::WORDS;
  for (uint32_t sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
    uint32_t lo = seg_list[3 * sg], hi = seg_list[3 * sg + 1], tlast = seg_list[3 * sg + 2];
    Jac<F> acc = Jac<F>::infinity();
    if (!IT::idle())
      for (uint32_t u = lo + IT::local(); u <= hi; u += PW) {
        const uint32_t* src = (u == tlast) ? piece_first : piece_last;
        acc = O::add(acc, MsmStored<GA>::load(src + (size_t)u * MsmStored<GA>::WORDS));
      }
    msm_wave_tree<G>(acc, my, partial + (size_t)sg * Jac<F>::WORDS);
  }
}
template <class G>
__global__ void __launch_bounds__(64) msm_big_bucket_kernel(const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count,
                                                            const uint32_t* __restrict__ partial, uint32_t* __restrict__ buckets,
                                                            uint32_t* __restrict__ scratch) {
  typedef MsmPairItems<G> IT;
  typedef MsmPairOps<G> O;
  typedef typename O::GA GA;
  typedef typename GA::F F;
  constexpr uint32_t PW = IT::PER_WAVE;
  const uint32_t nbig = big_count[0];
  uint32_t* my = scratch + (size_t)blockIdx.x * 64 * Jac<F>::WORDS;
  for (uint32_t b = blockIdx.x; b < nbig; b += gridDim.x) {
    uint32_t key = big_list[3 * b], s0 = big_list[3 * b + 1], ns = big_list[3 * b + 2];
    Jac<F> acc = Jac<F>::infinity();
    if (!IT::idle())
      for (uint32_t u = IT::local(); u < ns; u += PW) acc = O::add(acc, Jac<F>::load(partial + (size_t)(s0 + u) * Jac<F>::WORDS));
    msm_wave_tree<G>(acc, my, buckets + (size_t)key * Jac<F>::WORDS);
  }
}

// ------------------------------------------------------------------------------------------------ tail: sum_d d * B_d
// State per window: weighted array A (weights 1..mA) and plain array C;  V = WS(A) + S(C).
// One level with block size K = 2^k:  block j of A -> T_j (plain sum), L_j (weighted sum, weights 1..K)
//   V = S(C) + S(L) + WS({K * T_j}_{j>=1});   A' = {2^k T_j}_{j>=1},  C' = blocksums(C) ++ L
// (WIDE: two lanes per item where the group allows it -- pays when the level is latency-bound, i.e. has few items; a first level over
//  2^16 blocks fills the chip with one lane per item: G1-298 at c = 20 measured 0.71 -> 0.73 ms with two, G1-753 at c = 19 2.79 -> 2.32)
template <class G, bool WIDE>
__global__ void __launch_bounds__(64) msm_tail_level_kernel(const uint32_t* __restrict__ A_in, uint32_t mA, size_t strideA_in,
                                                            const uint32_t* __restrict__ C_in, uint32_t mC, size_t strideC_in,
                                                            uint32_t* __restrict__ A_out, size_t strideA_out,
                                                            uint32_t* __restrict__ C_out, size_t strideC_out, int k) {
  typedef MsmPairItems<G, WIDE> IT;
  typedef MsmPairOps<G, IT::TWO> O;
  typedef typename O::GA GA;
  typedef typename GA::F F;
  constexpr int PW = Jac<F>::WORDS;
  if (IT::idle()) return;
  uint32_t K = 1u << k;
  uint32_t JA = (mA + K - 1) >> k, JC = (mC + K - 1) >> k;
  uint32_t tid = IT::item();
  uint32_t w = blockIdx.y;
  if (tid < JA) {
    const uint32_t* A = A_in + w * strideA_in * PW;
    Jac<F> run = Jac<F>::infinity(), acc = Jac<F>::infinity();
    for (int tt = (int)K - 1; tt >= 0; tt--) {
      uint32_t i = tid * K + tt;
      if (i < mA) run = O::add(run, Jac<F>::load(A + (size_t)i * PW));
      acc = O::add(acc, run);
    }
    if (IT::writer()) acc.store(C_out + (w * strideC_out + JC + tid) * PW);
    if (tid >= 1) {
      for (int d = 0; d < k; d++) run = O::dbl(run);
      if (IT::writer()) run.store(A_out + (w * strideA_out + tid - 1) * PW);
    }
  } else if (tid < JA + JC) {
    uint32_t j = tid - JA;
    const uint32_t* Cw = C_in + w * strideC_in * PW;
    Jac<F> acc = Jac<F>::infinity();
    for (uint32_t tt = 0; tt < K; tt++) {
      uint32_t i = j * K + tt;
      if (i < mC) acc = O::add(acc, Jac<F>::load(Cw + (size_t)i * PW));
    }
    if (IT::writer()) acc.store(C_out + (w * strideC_out + j) * PW);
  }
}

// The same level for blocks of two, with the three dependent-free pieces of work on separate workgroups (blockIdx.z) so that
// the critical path of a level is one addition and one doubling instead of two additions and a doubling -- these levels
// are pure latency (a handful of lanes), so the extra doubling is free:
//   z = 0:  A'_{j-1} = 2 (A_{2j} + A_{2j+1})        z = 1:  L_j = A_{2j} + 2 A_{2j+1} -> C'[JC + j]        z = 2:  C'_j = C_{2j} + C_{2j+1}
// (items and operations: MsmPairItems / MsmPairOps above)
// one piece of a pair level: role 0 / 1 / 2 for item j of window w (the lanes of the item call it together)
template <class G, bool Q = true>
PCD_DEV void msm_tail_pair_item(int role, uint32_t j, uint32_t w, const uint32_t* A_in, uint32_t mA, size_t strideA_in, const uint32_t* C_in,
                                uint32_t mC, size_t strideC_in, uint32_t* A_out, size_t strideA_out, uint32_t* C_out, size_t strideC_out) {
  typedef MsmPairOps<G, MsmPairItems<G, true>::TWO, Q> O;
  typedef typename O::GA GA;
  typedef typename GA::F F;
  typedef MsmPairItems<G, true, Q> IT;
  constexpr int PW = Jac<F>::WORDS;
  const uint32_t JA = (mA + 1) >> 1, JC = (mC + 1) >> 1;
  if (role < 2) {
    if (j >= JA) return;
    const uint32_t* A = A_in + w * strideA_in * PW;
    const bool two = 2 * j + 1 < mA;
    const Jac<F> a0 = Jac<F>::load(A + (size_t)(2 * j) * PW);
    if (role == 0) {
      if (j == 0) return;
      Jac<F> run = two ? O::add(a0, Jac<F>::load(A + (size_t)(2 * j + 1) * PW)) : a0;
      run = O::dbl(run);
      if (IT::writer()) run.store(A_out + (w * strideA_out + j - 1) * PW);
    } else {
      Jac<F> acc = two ? O::add(O::dbl(Jac<F>::load(A + (size_t)(2 * j + 1) * PW)), a0) : a0;
      if (IT::writer()) acc.store(C_out + (w * strideC_out + JC + j) * PW);
    }
  } else {
    if (j >= JC) return;
    const uint32_t* Cw = C_in + w * strideC_in * PW;
    Jac<F> acc = Jac<F>::load(Cw + (size_t)(2 * j) * PW);
    if (2 * j + 1 < mC) acc = O::add(acc, Jac<F>::load(Cw + (size_t)(2 * j + 1) * PW));
    if (IT::writer()) acc.store(C_out + (w * strideC_out + j) * PW);
  }
}
template <class G, bool Q = true>
__global__ void __launch_bounds__(64) msm_tail_pair_kernel(const uint32_t* __restrict__ A_in, uint32_t mA, size_t strideA_in,
                                                           const uint32_t* __restrict__ C_in, uint32_t mC, size_t strideC_in,
                                                           uint32_t* __restrict__ A_out, size_t strideA_out,
                                                           uint32_t* __restrict__ C_out, size_t strideC_out) {
  typedef MsmPairItems<G, true, Q> IT;
  if (IT::idle()) return;
  msm_tail_pair_item<G, Q>((int)blockIdx.z, IT::item(), blockIdx.y, A_in, mA, strideA_in, C_in, mC, strideC_in, A_out, strideA_out, C_out, strideC_out);
}
// The LAST pair levels in one launch: once a level's 2 JA + JC pieces fit the item slots of one workgroup of FOUR waves -- one wave per
// SIMD of a CU: with eight, two waves share a SIMD and every level takes 1.6x as long, more than the launch it saves (measured: tail
// 0.66 -> 0.78 ms) -- a single workgroup per window runs every remaining level: the same pieces, the same ping-pong buffers
// (L2-resident by then), a workgroup barrier where the per-level kernels had a dependent launch (~8 us each, eight of them at c = 20).
// Not for the mailbox field variants (their LDS slots are per lane of ONE wave): the 753-bit groups keep a launch per level.
constexpr uint32_t MSM_FUSED_WAVES = 4;
// a pair level with more items than this (over all bucket windows; x 3 pieces of work each) keeps two lanes per operation for the prime-field
// groups: 4096 items x 3 x 4 lanes = 768 waves, under one wave per SIMD of the chip -- below that a level is pure latency and four lanes pay
#ifndef PCD_QUAD_MAX_ITEMS
#define PCD_QUAD_MAX_ITEMS 4096
#endif
constexpr uint32_t MSM_QUAD_MAX_ITEMS = PCD_QUAD_MAX_ITEMS;
template <class G>
struct MsmFusedTail {
  static constexpr bool ENABLED = !MsmPairOps<G>::GA::F::Base::MAILBOX;
  // the three kinds of pieces of a level (A' = 2 (A0 + A1), L = A0 + 2 A1, C' = C0 + C1) take WHOLE waves each: lanes of one wave that ran
  // different kinds would execute all three code paths one after the other
  static uint32_t waves(uint32_t items) { return (items + MsmPairItems<G>::PER_WAVE - 1) / MsmPairItems<G>::PER_WAVE; }
  static bool fits(uint32_t mA, uint32_t mC) { return ENABLED && 2 * waves((mA + 1) >> 1) + waves((mC + 1) >> 1) <= MSM_FUSED_WAVES; }
};
template <class G>
__global__ void __launch_bounds__(64 * MSM_FUSED_WAVES) msm_tail_fused_kernel(const uint32_t* A_in, uint32_t mA, size_t strideA_in, const uint32_t* C_in,
                                                                              uint32_t mC, size_t strideC_in, uint32_t* A0, uint32_t* A1, uint32_t* C0,
                                                                              uint32_t* C1, size_t strideAC, int flip) {
  typedef MsmPairItems<G> IT;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const bool idle = lane >= IT::PER_WAVE * IT::LANES;
  const uint32_t w = blockIdx.x;
  while (mA > 0 || mC > 1) {
    const uint32_t JA = (mA + 1) >> 1, JC = (mC + 1) >> 1;
    uint32_t* A_out = flip ? A1 : A0;
    uint32_t* C_out = flip ? C1 : C0;
    // waves [0, wa): kind 0, [wa, 2 wa): kind 1, the rest: kind 2 (uniform per wave)
    const uint32_t wa = (JA + IT::PER_WAVE - 1) / IT::PER_WAVE;
    const int role = wave < wa ? 0 : wave < 2 * wa ? 1 : 2;
    const uint32_t j = (wave - (uint32_t)role * wa) * IT::PER_WAVE + lane / IT::LANES;
    if (!idle) msm_tail_pair_item<G>(role, j, w, A_in, mA, strideA_in, C_in, mC, strideC_in, A_out, strideAC, C_out, strideAC);
    __syncthreads();  // (also makes this level's global stores visible to the whole workgroup)
    mA = JA ? JA - 1 : 0;
    mC = JC + JA;
    A_in = A_out; strideA_in = strideAC;
    C_in = C_out; strideC_in = strideAC;
    flip ^= 1;
  }
}

// total = sum_w 2^(c w) V_w, V_w = C[w * strideC];  plus `extra` points added at the end
template <class G>
__global__ void __launch_bounds__(64) msm_horner_kernel(const uint32_t* __restrict__ C, size_t strideC, int W, int c, uint32_t* __restrict__ out) {
  typedef typename MsmSingleItem<G>::GA GA;
  typedef typename GA::F F;
  typedef EC<GA> E;
  constexpr int PW = Jac<F>::WORDS;
  if (blockIdx.x != 0 || threadIdx.x >= MsmSingleItem<G>::LANES) return;
  Jac<F> total = Jac<F>::load(C + (size_t)(W - 1) * strideC * PW);
  for (int w = W - 2; w >= 0; w--) {
    for (int d = 0; d < c; d++) total = E::dbl(total);
    total = E::add(total, Jac<F>::load(C + (size_t)w * strideC * PW));
  }
  total.store(out);
}

// ------------------------------------------------------------------------------------------------ C-ABI <-> device image
template <class G>
__global__ void __launch_bounds__(64) points_abi_to_internal_kernel(const uint32_t* __restrict__ abi, uint32_t n, uint32_t* __restrict__ out) {
  typedef typename G::F F;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F>::from_abi(abi + (size_t)i * Aff<F>::ABI_WORDS).store(out + (size_t)i * MsmBaseStride<G>::value);
}
template <class G>
__global__ void __launch_bounds__(64) jac_internal_to_abi_kernel(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ abi) {
  typedef typename G::F F;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac<F> p = Jac<F>::load(in + (size_t)i * Jac<F>::WORDS);
  if (p.is_inf()) p = Jac<F>::infinity();  // canonical (0 : 1 : 0)
  p.to_abi(abi + (size_t)i * Jac<F>::ABI_WORDS);
}

// ------------------------------------------------------------------------------------------------ precomputed bases
// out[g * n + i] = 2^(shift * g) * P_i  (affine), g = 0 .. groups-1.  One lane per point; one-time cost at
// key upload (the proving key of a PCD is fixed for the whole computation).  The copies of one point are made in blocks of BLK: the
// doubling chain runs on in Jacobian form and the block shares ONE field inversion (Montgomery's trick inside the lane) -- the
// inversions, one per copy before, were five sixths of this kernel (753-bit: ~1 100 products each against ~200 for the doublings
// between two copies; key upload of BASELINE configs[2] 20.7 s).
template <class G>
__global__ void __launch_bounds__(64) msm_precompute_kernel(uint32_t* __restrict__ pts, uint32_t n, int groups, int shift) {
  // (one lane per point here: the 753-bit G1 computes in its mailbox variant, the extension-field groups in their plain form)
  typedef typename std::conditional<AccOf<G>::LANES == 1, typename AccOf<G>::type, G>::type GP;
  typedef typename GP::F F;
  typedef EC<GP> E;
  constexpr int BLK = (F::Base::N > 11 || F::DEG > 1) ? 4 : 8;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int STRIDE = MsmBaseStride<G>::value;
  Aff<F> p = Aff<F>::load(pts + (size_t)i * STRIDE);
  if (p.is_inf()) {  // every copy of the identity is the identity
    for (int g = 1; g < groups; g++) p.store(pts + ((size_t)g * n + i) * STRIDE);
    return;
  }
  Jac<F> q = {p.x, p.y, F::one()};
  for (int g0 = 1; g0 < groups; g0 += BLK) {
    const int nb = min(BLK, groups - g0);
    F zs[BLK], pre[BLK];
    unsigned infm = 0;  // (a point of odd prime order never doubles to the identity; kept for robustness)
#pragma unroll
    for (int j = 0; j < BLK; j++) {
      if (j < nb) {
        for (int d = 0; d < shift; d++) q = E::dbl(q);
        uint32_t* dst = pts + ((size_t)(g0 + j) * n + i) * STRIDE;
        q.X.store(dst);
        q.Y.store(dst + F::WORDS);
        const bool inf = q.is_inf();
        if (inf) infm |= 1u << j;
        zs[j] = inf ? F::one() : q.Z;
        pre[j] = j ? pre[j - 1] * zs[j] : zs[j];
      }
    }
    F inv = pre[0];
#pragma unroll
    for (int j = 1; j < BLK; j++) if (j == nb - 1) inv = pre[j];
    inv = inv.inv();
    Aff<F> last = {F::zero(), F::zero()};
#pragma unroll
    for (int j = BLK - 1; j >= 0; j--) {
      if (j < nb) {
        const F zi = j ? inv * pre[j - 1] : inv;
        inv = inv * zs[j];
        uint32_t* dst = pts + ((size_t)(g0 + j) * n + i) * STRIDE;
        const F zi2 = zi.sqr();
        Aff<F> a = {F::load(dst) * zi2, F::load(dst + F::WORDS) * zi2 * zi};
        if (infm & (1u << j)) a = {F::zero(), F::zero()};
        a.store(dst);
        if (j == nb - 1) last = a;
      }
    }
    q = (infm >> (nb - 1)) & 1u ? Jac<F>::infinity() : Jac<F>{last.x, last.y, F::one()};  // back to Z = 1: cheaper doublings, same point
  }
}
template <class G>
hipError_t msm_precompute(hipStream_t st, uint32_t* pts, uint32_t n, int groups, int shift) {
  if (n == 0 || groups <= 1) return hipSuccess;
  hipLaunchKernelGGL((msm_precompute_kernel<G>), dim3((n + 63) / 64), dim3(64), 0, st, pts, n, groups, shift);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ host driver
// The accumulate lane (round 5).  When several MSMs share the device -- the five of a Groth16 proof, the commitments of a Marlin round --
// their accumulate kernels go, one after the other, to ONE stream whose queue is confined to `cus` compute units by a CU mask
// (hipExtStreamCreateWithCUMask: the mask bits are dealt to the XCDs round-robin, tools/microbench/k3_cu_mask.hip, so clearing the last
// 8 k bits leaves k CUs free in every XCD), while everything latency-bound (sorts, fix-ups, bucket reductions, the assembly) stays on the
// MSM's own unmasked stream: an accumulate workgroup lives 0.5 .. 6 ms, and a dependent one-wave launch that has to wait for one of them to
// retire costs that much PER LAUNCH (measured: a chain of 20 one-wave launches 0.9 ms on an idle device, 20 .. 38 ms behind an unmasked
// grid of long-lived workgroups, 0.98 ms behind a masked one).  Accumulate grids are planned for the lane's CUs.
// MEASURED SLOWER than plain concurrency and off by default (pcdhip_groth16_set_schedule 2 turns it on): a masked queue runs the
// accumulate kernels 15 .. 45 % slower, and a proof already takes the sum of its kernels' standalone times (include/pcdhip.h).
struct MsmLane {
  hipStream_t stream = nullptr;
  int cus = 0;  // compute units the lane's queue may use
};

struct MsmWorkspace {
  // device buffers, grown on demand and reused across calls (no allocation on the hot path once warm)
  void* buf[28] = {nullptr};
  size_t cap[28] = {0};
  const MsmLane* lane = nullptr;                     // set by the caller around an msm_run whose accumulation goes to the lane
  hipEvent_t lane_in = nullptr, lane_out = nullptr;  // sorted list ready -> lane; accumulation done -> this MSM's own stream
  const uint32_t* last_err_dev = nullptr;  // device word raised by the last MSM's digit pass when a scalar was not reduced (null: not checked)
  int cus = 0;                             // compute units of the device this workspace lives on (queried on first use)
  hipError_t ensure(int slot, size_t bytes) {
    if (cap[slot] >= bytes) return hipSuccess;
    if (buf[slot]) { hipError_t e = hipFree(buf[slot]); if (e != hipSuccess) return e; buf[slot] = nullptr; cap[slot] = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&buf[slot], want);
    if (e != hipSuccess) return e;
    cap[slot] = want;
    return hipSuccess;
  }
  void release() {
    for (int i = 0; i < 28; i++) if (buf[i]) { (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0; }
    if (lane_in) { (void)hipEventDestroy(lane_in); lane_in = nullptr; }
    if (lane_out) { (void)hipEventDestroy(lane_out); lane_out = nullptr; }
  }
};

struct MsmTimings {  // milliseconds, filled when events are requested
  float digits = 0, scan = 0, scatter = 0, accumulate = 0, fixup = 0, tail = 0, horner = 0, total = 0;
  uint32_t entries = 0, chunk = 0;  // the sorted list's length and the entries per lane the device chose for it (pcdhip_msm_last_plan)
};

enum { WS_CNT = 0, WS_OFF, WS_BSUM, WS_SORTED, WS_BUCKETS, WS_PFIRST, WS_PLAST, WS_BIG, WS_BIGSCR, WS_A0, WS_A1, WS_C0, WS_C1, WS_OUT, WS_SCAL,
       WS_BIGPART, WS_ONES, WS_SLOTS, WS_CUR, WS_ENTRIES, WS_ACCB, WS_TLIST, WS_TPAIR, WS_TPREFIX, WS_TPTS };

#define PCD_HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// scalars_dev: n * NS u32 (canonical), bases_dev: n affine points;  out_dev: one Jacobian point.
struct MsmBasesView {
  const uint32_t* dptr;  // groups * n_total affine points: group g holds 2^(c * Wg * g) P_i
  uint32_t n_total;      // points per group
  uint32_t offset;       // first point of this MSM
  int c;                 // window bits the groups were built for (0: no precomputation, free choice)
  int groups;            // 1: no precomputation
  const uint32_t* inf_bits = nullptr;  // bit i set: point i (of n_total) is the point at infinity; null: none is (msm_base_is_inf)
};

// Several MSMs over the SAME scalar vector and identically laid-out base arrays (the a, b_g1, b_g2 and l queries of one
// Groth16 proof all take the assignment z) need the digit extraction and the sort of the (bucket, base index) entries
// only once: the producer publishes its sorted list, the consumers wait for `ready` and start at the accumulation.
struct MsmSharedSort {
  bool valid = false;
  uint32_t n = 0, n_total = 0, offset = 0, tkeys = 0;
  int c = 0, W = 0, groups = 0;
  const uint32_t* scalars = nullptr;
  const uint32_t* inf_bits = nullptr;  // the bitmap the producer filtered its list with (null: the list holds every entry)
  MsmEntrySource src{};
  const uint32_t* off = nullptr;
  hipEvent_t ready = nullptr;  // created by the caller (timing disabled); recorded by the producer after the sort
};
enum { MSM_SHARE_NONE = 0, MSM_SHARE_PRODUCE = 1, MSM_SHARE_CONSUME = 2 };

template <class G>
hipError_t msm_run(MsmWorkspace& ws, hipStream_t st, const MsmBasesView& bv, const uint32_t* scalars_dev, uint32_t n,
                   uint32_t* out_dev, int c_override, uint32_t chunk_override, int sort_mode, MsmTimings* tm,
                   MsmSharedSort* share = nullptr, int share_role = MSM_SHARE_NONE) {
  // sort_mode: bits 0-3 the sort (pcdhip_msm_set_sort), bits 4-5 the accumulation form (pcdhip_msm_set_accumulate: 0 by size, 1 running
  // sums, 2 pair tree), bits 8-19 / 20-27 the pair tree's chunk and smallest batch when given
  const int acc_mode = (sort_mode >> 4) & 3;
  const uint32_t tree_chunk = ((uint32_t)sort_mode >> 8) & 0xFFFu, tree_min_pairs = ((uint32_t)sort_mode >> 20) & 0xFFu;
  sort_mode &= 15;
  const bool single_pass = sort_mode == 1;  // pcdhip_msm_set_sort(ctx, 1)
  const uint32_t* bases_dev = bv.dptr;
  typedef typename G::F F;
  constexpr int NS = G::FR::N32;  // canonical scalar words
  constexpr int PW = Jac<F>::WORDS;
  constexpr size_t PB = (size_t)PW * 4;
  if (n == 0) { ws.last_err_dev = nullptr; Jac<F> inf = Jac<F>::infinity(); return hipMemcpyAsync(out_dev, &inf, PB, hipMemcpyHostToDevice, st); }
  MsmPlan pl;
  pl.n = n;
  pl.c = bv.groups > 1 ? bv.c : (c_override ? c_override : msm_pick_window(n, G::FR::BITS, 0));
  pl.W = msm_num_windows(G::FR::BITS, pl.c);
  const int Wg = (pl.W + bv.groups - 1) / bv.groups;  // bucket windows
  pl.nkeys = (uint32_t)Wg << pl.c;
  const uint64_t maxM = (uint64_t)n * pl.W;
  // sorted entries per lane (median-of-8 A/B on MI355X at 2^18 / 2^20, G1 and G2: 40 beats 32 by 3-5 %, 48 ties, 64 loses) -- nudged, among
  // 40 .. 56, to the value whose wave count fills whole rounds of the chip: a one-wave-per-SIMD kernel that needs 14.4 rounds idles
  // 60 % of the SIMDs during the last one (G1-753 at 2^20: 48 entries = 12.0 rounds, 55.5 -> 53.7 ms)
  pl.chunk = chunk_override ? chunk_override : 40;
  int& cus = ws.cus;  // cached per workspace (= per context and stream: no sharing between host threads, right device)
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  // the pair-tree form of the accumulation (msm_pair_tree_kernel): long lists of the 753-bit G1 groups.  One chunk per lane of the
  // resident waves and as many rounds as keep a chunk below MSM_TREE_CHUNK_MAX entries (the scratch areas grow with the chunk)
  bool use_tree = false;
  uint32_t tree_cap = 0;
  const uint32_t* tree_chunk_dev = nullptr;
  if constexpr (MsmTreeCapable<G>::value) {
    use_tree = acc_mode == 2 && (uint64_t)bv.n_total * bv.groups < MSM_SCR;
    if (use_tree) {
      // (host side: the CAPACITY of a chunk; the chunk itself is chosen on the device, from the actual list length)
      tree_cap = tree_chunk ? tree_chunk : MSM_TREE_CHUNK_MAX;
      pl.chunk = tree_cap;
    }
  }
  const MsmLane* lane = ws.lane;
  const int plan_cus = lane && lane->cus > 0 ? lane->cus : cus;  // the accumulate grid fills whole rounds of the CUs its queue may use
  // (round 5: the chunk is chosen on the device from the actual list length -- msm_plan_chunk -- between chunk_lo and 56; the host
  //  sizes grids and piece arrays for chunk_lo, which is 40 as before for a list that is long even when sparse and goes down to 16 for one
  //  that cannot fill four rounds of the lanes)
  uint32_t acc_lanes = 0, chunk_lo = 0;
  static const uint32_t chunk_hi_env = getenv("PCDHIP_CHUNK_HI") ? (uint32_t)atoi(getenv("PCDHIP_CHUNK_HI")) : 0u;  // developer knob (A/B of the upper bound)
  const uint32_t chunk_hi = chunk_hi_env >= 16 && chunk_hi_env <= 63 ? chunk_hi_env : 56;
  if (!chunk_override && !use_tree) {
    acc_lanes = (uint32_t)plan_cus * 4 * MsmAccWaves<G>::value * (64 / AccOf<G>::LANES);
    chunk_lo = (uint32_t)std::min<uint64_t>(40, std::max<uint64_t>(16, (maxM + 4ull * acc_lanes - 1) / (4ull * acc_lanes)));
    pl.chunk = chunk_lo;
  }
  const uint32_t plan_word = acc_lanes && acc_lanes < (1u << 20) && chunk_hi < 64 ? msm_plan_word(acc_lanes, chunk_lo, chunk_hi) : 0u;
  if (acc_lanes && !plan_word) pl.chunk = 40;  // (a device or knob outside the plan word's ranges: the fixed chunk of rounds 1-4)
  if (maxM >= 0xFFFFFFF0ull || (uint64_t)bv.n_total * bv.groups >= 0x7FFFFFF0ull) return hipErrorInvalidValue;  // bit 31 of an entry: sign

  EventSet<9> ev;
  if (tm) PCD_HIP_TRY(ev.create());
  auto mark = [&](int i) -> hipError_t { return tm ? hipEventRecord(ev[i], st) : hipSuccess; };

  // keys: pl.nkeys real buckets + 1 pseudo bucket (scalars equal to one)
  const uint32_t tkeys = pl.nkeys + 1, ones_key = pl.nkeys;
  const bool consume = share && share_role == MSM_SHARE_CONSUME && share->valid && share->n == n && share->n_total == bv.n_total &&
                       share->offset == bv.offset && share->tkeys == tkeys && share->c == pl.c && share->W == pl.W &&
                       share->groups == bv.groups && share->scalars == scalars_dev &&
                       // a list filtered by a bitmap lacks the entries of its flagged bases: only a consumer that flags the SAME bases may take
                       // it (an unfiltered list serves everyone: entries on bases at infinity add the identity); otherwise this MSM sorts for itself
                       (share->inf_bits == nullptr || share->inf_bits == bv.inf_bits);
  PCD_HIP_TRY(ws.ensure(WS_CNT, (size_t)tkeys * 4 + 16));
  PCD_HIP_TRY(ws.ensure(WS_OFF, ((size_t)tkeys + 1) * 4));
  const uint32_t scan_per_block = 16384;
  const uint32_t scan_blocks = (tkeys + scan_per_block - 1) / scan_per_block;
  PCD_HIP_TRY(ws.ensure(WS_BSUM, (size_t)scan_blocks * 4));
  PCD_HIP_TRY(ws.ensure(WS_SORTED, (size_t)maxM * 4));
  PCD_HIP_TRY(ws.ensure(WS_BUCKETS, (size_t)tkeys * PB));
  PCD_HIP_TRY(ws.ensure(WS_OUT, PB + 64));
  PCD_HIP_TRY(ws.ensure(WS_ONES, (size_t)n * 4 + 32));
  uint32_t* cnt = (uint32_t*)ws.buf[WS_CNT];
  uint32_t* off = (uint32_t*)ws.buf[WS_OFF];
  uint32_t* bsum = (uint32_t*)ws.buf[WS_BSUM];
  uint32_t* sorted = (uint32_t*)ws.buf[WS_SORTED];
  uint32_t* buckets = (uint32_t*)ws.buf[WS_BUCKETS];
  uint32_t* ones_idx = (uint32_t*)ws.buf[WS_ONES];
  uint32_t* flag = ones_idx + n;  // cap-overflow flag of the single-pass binning
  uint32_t* err = flag + 1;       // a scalar >= 2^bits was seen (msm_scalar_too_wide)
  uint32_t* big_count = flag + 2; // [#big buckets, #segments] of the fix-up pass; flag + 4: the pair tree's chunk word
  auto zero_start = [&](uint32_t n_cnt) {  // cnt[0 .. n_cnt) and the four control words, one launch
    hipLaunchKernelGGL(msm_zero_kernel, dim3(std::max<uint32_t>(1u, (n_cnt + 255) / 256)), dim3(256), 0, st, cnt, n_cnt, flag);
  };
  ws.last_err_dev = consume ? nullptr : err;
  constexpr int SBITS = G::FR::BITS;
  // Optional single-pass binning: every bucket owns `cap` slots (mean load + 6 sigma + 8), one atomic pass instead of
  // histogram + scatter, on-device fallback to the compact list when a bucket overflows.  Measured on MI355X at
  // n = 2^20: it wins for witness-like scalars (0.37 vs 0.52 ms of sorting) and loses for uniform ones (1.64 vs 1.45 ms:
  // 4-byte writes scattered over a 200 MB slot array against a 63 MB compact list), so the two-pass counting sort
  // stays the default (pcdhip_msm_set_sort).
  const int wins = (pl.W + Wg - 1) / Wg;  // scalar windows that share one bucket window
  const int top_bits = G::FR::BITS + 1 - (pl.W - 1) * pl.c;
  double mu = (double)n * wins / (double)(1u << (pl.c - 1)) + (top_bits < pl.c ? (double)n / (double)(1u << top_bits) : 0.0);
  uint32_t cap = (uint32_t)(mu + 6.0 * sqrt(mu) + 8.0);
  cap = (cap + 7u) & ~7u;
  const bool use_slots = single_pass && (double)pl.nkeys * cap * 4.0 <= 1.5e9;
  // default: MSD partition through LDS (needs whole bins of 2^MSM_BIN_SHIFT keys and an LDS histogram per workgroup)
  const uint32_t nbins = pl.nkeys >> MSM_BIN_SHIFT;
  const bool use_partition = sort_mode == 0 && pl.c >= MSM_BIN_SHIFT && nbins >= 1 && nbins <= MSM_MAX_BINS && n >= 4096;
  uint32_t* slots = nullptr;
  if (use_slots) { PCD_HIP_TRY(ws.ensure(WS_SLOTS, (size_t)pl.nkeys * cap * 4)); slots = (uint32_t*)ws.buf[WS_SLOTS]; }

  PCD_HIP_TRY(mark(0));
  dim3 gd((n + 255) / 256), bd(256);
  MsmEntrySource src;
  if (consume) {
    // the producer's sorted list (its workspace is not touched again before every consumer has finished)
    PCD_HIP_TRY(hipStreamWaitEvent(st, share->ready, 0));
    zero_start(0);
    src = share->src;
    off = const_cast<uint32_t*>(share->off);
    PCD_HIP_TRY(mark(1)); PCD_HIP_TRY(mark(2)); PCD_HIP_TRY(mark(3));
  } else if (use_slots) {
    zero_start(tkeys);
    // 1. one pass: slots + exact histogram
    hipLaunchKernelGGL((msm_digits_kernel<NS, MODE_BIN>), gd, bd, 0, st, scalars_dev, n, pl.c, pl.W, Wg, bv.n_total, bv.offset, pl.nkeys, cnt,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, slots, cap, ones_idx, flag, 0, SBITS, err, bv.inf_bits);
    PCD_HIP_TRY(mark(1));
    // 2. scan
    hipLaunchKernelGGL(scan_block_sums, dim3(scan_blocks), dim3(1024), 0, st, cnt, tkeys, scan_per_block, bsum);
    hipLaunchKernelGGL(scan_apply, dim3(scan_blocks), dim3(1024), 0, st, cnt, tkeys, scan_per_block, bsum, scan_blocks, off);
    PCD_HIP_TRY(mark(2));
    // 3. fallback compaction: runs (on the device's own decision) only if some bucket overflowed its slots
    PCD_HIP_TRY(ws.ensure(WS_CUR, (size_t)tkeys * 4));
    uint32_t* cur = (uint32_t*)ws.buf[WS_CUR];
    PCD_HIP_TRY(hipMemsetAsync(cur, 0, (size_t)tkeys * 4, st));
    hipLaunchKernelGGL((msm_digits_kernel<NS, MODE_SCATTER>), gd, bd, 0, st, scalars_dev, n, pl.c, pl.W, Wg, bv.n_total, bv.offset, pl.nkeys, cur,
                       off, sorted, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr, flag, 1, SBITS, err, bv.inf_bits);
    PCD_HIP_TRY(mark(3));
    src = {sorted, slots, ones_idx, flag, cap, ones_key};
  } else if (use_partition) {
    zero_start(nbins + 1);  // (this path's histogram is per BIN)
    // 1. coarse histogram  2. bin bases  3. partition + per-bin counting sort (see "partition sort" above)
    const uint32_t tiles = (n + MSM_TILE - 1) / MSM_TILE;
    PCD_HIP_TRY(ws.ensure(WS_ENTRIES, (size_t)maxM * 8));
    PCD_HIP_TRY(ws.ensure(WS_CUR, ((size_t)2 * nbins + 8) * 4));
    uint64_t* entries = (uint64_t*)ws.buf[WS_ENTRIES];
    uint32_t* bin_base = (uint32_t*)ws.buf[WS_CUR];
    uint32_t* cursor = bin_base + nbins + 2;
    uint32_t* ones_count = flag;  // (the binning flag word is unused on this path)
    hipLaunchKernelGGL((msm_coarse_kernel<NS, false>), dim3(tiles), dim3(256), (size_t)nbins * 4, st, scalars_dev, n, pl.c, pl.W, Wg, bv.n_total,
                       bv.offset, nbins, cnt, (uint64_t*)nullptr, ones_count, ones_idx, SBITS, err, bv.inf_bits);
    PCD_HIP_TRY(mark(1));
    hipLaunchKernelGGL(msm_bin_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, nbins, bin_base, cursor, ones_count, off, pl.nkeys);
    PCD_HIP_TRY(mark(2));
    hipLaunchKernelGGL((msm_coarse_kernel<NS, true>), dim3(tiles), dim3(256), (size_t)nbins * 4, st, scalars_dev, n, pl.c, pl.W, Wg, bv.n_total,
                       bv.offset, nbins, cursor, entries, ones_count, ones_idx, SBITS, err, bv.inf_bits);
    hipLaunchKernelGGL(msm_bin_sort_kernel, dim3(nbins), dim3(256), 0, st, entries, bin_base, sorted, off);
    PCD_HIP_TRY(mark(3));
    src = {sorted, nullptr, ones_idx, nullptr, 0u, ones_key};
  } else {
    zero_start(tkeys);
    // 1. histogram  2. scan  3. scatter (cursor = cnt reset to zero)
    hipLaunchKernelGGL((msm_digits_kernel<NS, MODE_HIST>), gd, bd, 0, st, scalars_dev, n, pl.c, pl.W, Wg, bv.n_total, bv.offset, pl.nkeys, cnt,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, 0, SBITS, err, bv.inf_bits);
    PCD_HIP_TRY(mark(1));
    hipLaunchKernelGGL(scan_block_sums, dim3(scan_blocks), dim3(1024), 0, st, cnt, tkeys, scan_per_block, bsum);
    hipLaunchKernelGGL(scan_apply, dim3(scan_blocks), dim3(1024), 0, st, cnt, tkeys, scan_per_block, bsum, scan_blocks, off);
    PCD_HIP_TRY(mark(2));
    PCD_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)tkeys * 4, st));
    hipLaunchKernelGGL((msm_digits_kernel<NS, MODE_SCATTER>), gd, bd, 0, st, scalars_dev, n, pl.c, pl.W, Wg, bv.n_total, bv.offset, pl.nkeys, cnt,
                       off, sorted, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, 0, SBITS, err, bv.inf_bits);
    PCD_HIP_TRY(mark(3));
    src = {sorted, nullptr, nullptr, nullptr, 0u, ones_key};
  }
  if (share && share_role == MSM_SHARE_PRODUCE) {
    share->valid = true;
    share->n = n; share->n_total = bv.n_total; share->offset = bv.offset; share->tkeys = tkeys;
    share->c = pl.c; share->W = pl.W; share->groups = bv.groups; share->scalars = scalars_dev; share->inf_bits = bv.inf_bits;
    share->src = src; share->off = off;
    PCD_HIP_TRY(hipEventRecord(share->ready, st));
  }
  // 4. accumulate.  Nothing below waits for the host: grids are sized for the largest possible list (n W entries;
  //    the actual count M = off[nkeys] is read on the device) so the whole MSM is one asynchronous chain of launches.
  uint32_t nchunks = (uint32_t)((maxM + pl.chunk - 1) / pl.chunk);
  uint32_t tree_grid = 0;
  if (use_tree) {
    // persistent waves; at most lanes x rounds chunks whatever the device picks (msm_tree_plan_kernel), or the forced chunk's count
    // (every SIMD gets a wave as soon as the list allows chunks of 32: a mid-size list is spread thin rather than over few waves)
    tree_grid = std::min<uint32_t>(tree_chunk ? (nchunks + 63) / 64 : (uint32_t)((maxM / 32 + 63) / 64), (uint32_t)plan_cus * 4);
    if (!tree_chunk) nchunks = tree_grid * 64 * (uint32_t)((maxM + (uint64_t)tree_grid * 64 * tree_cap - 1) / ((uint64_t)tree_grid * 64 * tree_cap));
  }
  typedef MsmStored<typename SplitOf<G>::type> Stored;
  constexpr size_t RB = (size_t)Stored::WORDS * 4;  // bytes of a flushed record
  PCD_HIP_TRY(ws.ensure(WS_PFIRST, (size_t)nchunks * RB));
  PCD_HIP_TRY(ws.ensure(WS_PLAST, (size_t)nchunks * RB));
  uint32_t* acc_buckets = buckets;  // where whole-bucket flushes go: the bucket array itself, or (wider records) an array of their own
  if (Stored::SEPARATE) { PCD_HIP_TRY(ws.ensure(WS_ACCB, (size_t)tkeys * RB)); acc_buckets = (uint32_t*)ws.buf[WS_ACCB]; }
  const uint32_t big_limit = 8, seg_len = 128;  // pieces per segment: 4 per item and a 5-level tree with two lanes per addition
  const uint32_t big_cap = nchunks / big_limit + 2;   // a big bucket spans more than big_limit chunks
  const uint32_t seg_cap = nchunks / seg_len + big_cap + 2;
  PCD_HIP_TRY(ws.ensure(WS_BIG, (size_t)(3 * big_cap + 3 * seg_cap + 8) * 4));
  PCD_HIP_TRY(ws.ensure(WS_BIGPART, (size_t)seg_cap * PB));
  uint32_t* pfirst = (uint32_t*)ws.buf[WS_PFIRST];
  uint32_t* plast = (uint32_t*)ws.buf[WS_PLAST];
  uint32_t* big = (uint32_t*)ws.buf[WS_BIG];
  uint32_t* seg_list = big + 3 * big_cap;
  uint32_t* big_partial = (uint32_t*)ws.buf[WS_BIGPART];
  // (no memset of the bucket array -- 69 MB at c = 19: every bucket is written by exactly one of msm_accumulate, msm_fixup
  //  (also the empty ones: Z = 0) and msm_big_bucket)
  PCD_HIP_TRY(mark(8));  // the accumulate stage time is the kernel alone (mark 8 -> mark 4)
  hipStream_t own = st;
  if (lane) {  // the accumulation runs on the lane, behind the other MSMs' accumulations queued there before it
    if (!ws.lane_in) PCD_HIP_TRY(hipEventCreateWithFlags(&ws.lane_in, hipEventDisableTiming));
    if (!ws.lane_out) PCD_HIP_TRY(hipEventCreateWithFlags(&ws.lane_out, hipEventDisableTiming));
    PCD_HIP_TRY(hipEventRecord(ws.lane_in, own));
    PCD_HIP_TRY(hipStreamWaitEvent(lane->stream, ws.lane_in, 0));
    st = lane->stream;
  }
  {
    constexpr uint32_t per_wave = 64 / AccOf<G>::LANES;  // chunks per workgroup
    const dim3 acc_grid((nchunks + per_wave - 1) / per_wave);
    // (with per-bucket slots both instantiations are queued: the device's overflow flag decides which of them does the work)
    bool launched = false;
    if constexpr (MsmTreeCapable<G>::value) {
      if (use_tree) {
        constexpr int FW = AccOf<G>::type::F::WORDS;
        const uint32_t grid = tree_grid;
        MsmTreeBufs tb;
        tb.cap = tree_cap;
        tb.phases = 3;
        uint32_t* chunk_word = big_count + 2;
        hipLaunchKernelGGL(msm_tree_plan_kernel, dim3(1), dim3(1), 0, st, off, tkeys, grid * 64, tree_cap, tree_chunk, chunk_word);
        tree_chunk_dev = chunk_word;
        PCD_HIP_TRY(ws.ensure(WS_TLIST, MsmTreeBufs::list_words(tb.cap) * 4 * grid));
        PCD_HIP_TRY(ws.ensure(WS_TPAIR, MsmTreeBufs::pair_words(tb.cap) * 4 * grid));
        PCD_HIP_TRY(ws.ensure(WS_TPREFIX, MsmTreeBufs::prefix_words(tb.cap, FW) * 4 * grid));
        PCD_HIP_TRY(ws.ensure(WS_TPTS, MsmTreeBufs::pts_words(tb.cap, FW) * 4 * grid));
        tb.list = (uint32_t*)ws.buf[WS_TLIST]; tb.pairs = (uint32_t*)ws.buf[WS_TPAIR];
        tb.prefix = (uint32_t*)ws.buf[WS_TPREFIX]; tb.pts = (uint32_t*)ws.buf[WS_TPTS];
        hipLaunchKernelGGL((msm_pair_tree_kernel<G>), dim3(grid), dim3(64), 0, st, bases_dev, src, off, tkeys, chunk_word, tb,
                           tree_min_pairs ? tree_min_pairs : MSM_TREE_MIN_PAIRS, acc_buckets, pfirst, plast);
        launched = true;
      }
    }
    if (!launched) {
      if (src.slots) hipLaunchKernelGGL((msm_accumulate_kernel<G, false>), acc_grid, dim3(64), 0, st, bases_dev, src, off, tkeys, pl.chunk, acc_buckets, pfirst, plast, plan_word);
      hipLaunchKernelGGL((msm_accumulate_kernel<G, true>), acc_grid, dim3(64), 0, st, bases_dev, src, off, tkeys, pl.chunk, acc_buckets, pfirst, plast, plan_word);
    }
  }
  if (lane) {
    PCD_HIP_TRY(hipEventRecord(ws.lane_out, st));
    st = own;
    PCD_HIP_TRY(hipStreamWaitEvent(st, ws.lane_out, 0));
  }
  PCD_HIP_TRY(mark(4));
  // 5. pieces
  hipLaunchKernelGGL((msm_fixup_kernel<G>), dim3(MsmItems<G>::grid(tkeys)), dim3(64), 0, st, off, tkeys, pl.chunk, pfirst, plast, acc_buckets, buckets,
                     big_limit, big_count, big, big_cap, seg_list, seg_len, tree_chunk_dev, plan_word);
  {
    const uint32_t big_grid = std::min<uint32_t>(seg_cap, 2048);
    PCD_HIP_TRY(ws.ensure(WS_BIGSCR, (size_t)big_grid * 64 * PB));
    hipLaunchKernelGGL((msm_big_segments_kernel<G>), dim3(big_grid), dim3(64), 0, st, seg_list, big_count, pfirst, plast, big_partial,
                       (uint32_t*)ws.buf[WS_BIGSCR]);
    hipLaunchKernelGGL((msm_big_bucket_kernel<G>), dim3(std::min<uint32_t>(big_cap, 2048)), dim3(64), 0, st, big, big_count, big_partial,
                       buckets, (uint32_t*)ws.buf[WS_BIGSCR]);
  }
  hipLaunchKernelGGL((msm_merge_ones_kernel<G>), dim3(1), dim3(64), 0, st, buckets, ones_key);
  PCD_HIP_TRY(mark(5));
  // 6. tail levels
  {
    const uint32_t B = 1u << (pl.c - 1);  // buckets |d| = 1 .. 2^(c-1) of a window (signed digits)
    size_t cap_pts = ((size_t)1 << pl.c);  // generous per-window capacity for A'/C'
    PCD_HIP_TRY(ws.ensure(WS_A0, (size_t)Wg * cap_pts * PB / 2 + PB * Wg * 4));
    PCD_HIP_TRY(ws.ensure(WS_A1, (size_t)Wg * cap_pts * PB / 2 + PB * Wg * 4));
    PCD_HIP_TRY(ws.ensure(WS_C0, (size_t)Wg * cap_pts * PB / 2 + PB * Wg * 4));
    PCD_HIP_TRY(ws.ensure(WS_C1, (size_t)Wg * cap_pts * PB / 2 + PB * Wg * 4));
    const size_t strideAC = cap_pts / 2 + 4;
    const uint32_t* A_in = buckets + PW;  // bucket d = 1 of window 0
    size_t strideA_in = (size_t)1 << pl.c;
    const uint32_t* C_in = nullptr;
    size_t strideC_in = 0;
    uint32_t mA = B, mC = 0;
    int flip = 0, level = 0;
    bool fused = false;
    while (mA > 0 || mC > 1) {
      // blocks of 8 at the first (throughput-bound) level, then pairs: the later levels are pure latency, and per halving
      // of the array a pair level costs 2 additions + 1 doubling against 3.6 / 5.3 addition-equivalents for blocks of 4 / 8
      // (the first level of a large bucket array is throughput-bound: blocks of 8 there)
      int k = (level == 0 && B >= (1u << 17)) ? 3 : 1;
      uint32_t K = 1u << k;
      uint32_t JA = (mA + K - 1) >> k, JC = (mC + K - 1) >> k;
      uint32_t* A_out = (uint32_t*)ws.buf[flip ? WS_A1 : WS_A0];
      uint32_t* C_out = (uint32_t*)ws.buf[flip ? WS_C1 : WS_C0];
      uint32_t threads = JA + JC;
      if (!fused && level > 0 && MsmFusedTail<G>::fits(mA, mC)) {  // every remaining level in one workgroup per window
        hipLaunchKernelGGL((msm_tail_fused_kernel<G>), dim3(Wg), dim3(64 * MSM_FUSED_WAVES), 0, st, A_in, mA, strideA_in, C_in, mC, strideC_in,
                           (uint32_t*)ws.buf[WS_A0], (uint32_t*)ws.buf[WS_A1], (uint32_t*)ws.buf[WS_C0], (uint32_t*)ws.buf[WS_C1], strideAC, flip);
        fused = true;
      }
      if (fused) {
        // (the host only follows the level sequence to know where the result ends up)
      } else if (k == 1 && std::max(JA, JC) * (uint32_t)Wg > MSM_QUAD_MAX_ITEMS)   // a level this wide is throughput: two lanes per operation
        hipLaunchKernelGGL((msm_tail_pair_kernel<G, false>), dim3(MsmPairItems<G, true, false>::grid(std::max(JA, JC)), Wg, 3), dim3(64), 0, st, A_in, mA,
                           strideA_in, C_in, mC, strideC_in, A_out, strideAC, C_out, strideAC);
      else if (k == 1)
        hipLaunchKernelGGL((msm_tail_pair_kernel<G>), dim3(MsmPairItems<G>::grid(std::max(JA, JC)), Wg, 3), dim3(64), 0, st, A_in, mA, strideA_in, C_in,
                           mC, strideC_in, A_out, strideAC, C_out, strideAC);
      else if (threads <= (1u << 15))
        hipLaunchKernelGGL((msm_tail_level_kernel<G, true>), dim3(MsmPairItems<G, true>::grid(threads), Wg), dim3(64), 0, st, A_in, mA, strideA_in, C_in, mC,
                           strideC_in, A_out, strideAC, C_out, strideAC, k);
      else
        hipLaunchKernelGGL((msm_tail_level_kernel<G, false>), dim3(MsmPairItems<G, false>::grid(threads), Wg), dim3(64), 0, st, A_in, mA, strideA_in, C_in, mC,
                           strideC_in, A_out, strideAC, C_out, strideAC, k);
      mA = JA ? JA - 1 : 0;
      mC = JC + JA;
      A_in = A_out; strideA_in = strideAC;
      C_in = C_out; strideC_in = strideAC;
      flip ^= 1;
      level++;
    }
    PCD_HIP_TRY(mark(6));
    // 7. windows
    hipLaunchKernelGGL((msm_horner_kernel<G>), dim3(1), dim3(64), 0, st, C_in, strideC_in, Wg, pl.c, out_dev);
    PCD_HIP_TRY(mark(7));
  }
  PCD_HIP_TRY(hipGetLastError());
  if (tm) {
    PCD_HIP_TRY(hipStreamSynchronize(st));
    auto el = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[a], ev[b]); return ms; };
    tm->digits = el(0, 1); tm->scan = el(1, 2); tm->scatter = el(2, 8); tm->accumulate = el(8, 4);
    tm->fixup = el(4, 5); tm->tail = el(5, 6); tm->horner = el(6, 7); tm->total = el(0, 7);
    PCD_HIP_TRY(hipMemcpy(&tm->entries, off + tkeys, 4, hipMemcpyDeviceToHost));
    tm->chunk = plan_word ? msm_chunk_of_plan(tm->entries, plan_word) : pl.chunk;
    if (tree_chunk_dev) PCD_HIP_TRY(hipMemcpy(&tm->chunk, tree_chunk_dev, 4, hipMemcpyDeviceToHost));
  }
  return hipSuccess;
}

}  // namespace pcd
