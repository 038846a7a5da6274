// TEST HARNESS ONLY (never part of libpcdhip.so): the LDS-mailbox variants of the lane-split 753-bit extension fields against the plain
// lane-split variants, operation by operation, on the GPU -- field product / square and the group operations the MSM kernels are made
// of (madd, add, dbl) -- under RANDOM ACTIVE-ITEM MASKS and DIVERGENT branches: per round an item (a lane pair / triple) may sit out,
// take the doubling path of an addition (equal points), the cancellation path (opposite points), an infinity operand, or a branch with
// a different operation mix than its neighbours.  Both forms must give the same VALUES (representatives in [0, 2p) may differ).
// Also the round-3 reproducer: one item, three active lanes, both operands finite, result compared AS STORED after canonicalisation on
// the device (the shape in which ROCm 7.2's MachineCopyPropagation dropped a live copy; profiles/DESIGN_history_r01-r05.md section 4).
// Built by __graft_entry__.build() (hipcc, gfx950) into tests/gpucheck/libgpucheck.so; tests/test_gpu_mailbox.py drives it.
#include <cstdio>
#include <vector>
#include "../../pcd_amd/csrc/common.h"
using namespace pcd;

template <class FD, class FS> __device__ FD conv(const FS& a) { FD r; for (int i = 0; i < FS::N; i++) r.v[i] = a.v[i]; return r; }
template <class B> __device__ B rnd(uint32_t& s) {
  B r;
  for (int i = 0; i < B::N; i++) { s = s * 1664525u + 1013904223u; r.v[i] = (s >> 4) & 0x0FFFFFFFu; }
  r.v[B::N - 1] &= 0xFFFFu;  // < 2^(28 (N-1) + 16) < p
  return r;
}
template <class B> __device__ bool same(const B& a0, const B& b0) {  // equal mod p
  const B a = a0.canonical(), b = b0.canonical();
  bool ok = true;
  for (int i = 0; i < B::N; i++) ok &= a.v[i] == b.v[i];
  return ok;
}
template <class FS, class FM> __device__ bool same_pt(const Jac<FM>& b, const Jac<FS>& a) {
  typedef typename FS::Base BS;
  // compare as group elements of the same projective class is too weak here: both forms run the SAME formulas, so coordinates agree
  return same(conv<BS>(b.X.c), a.X.c) && same(conv<BS>(b.Y.c), a.Y.c) && same(conv<BS>(b.Z.c), a.Z.c);
}

// GS: plain split config, GM: mailbox split config (same lane layout).  bad[k] counts mismatches of check k.
template <class GS, class GM>
__global__ void __launch_bounds__(64) check(uint32_t* bad, int rounds, int lanes_used, uint32_t seed) {
  typedef typename GS::F FS; typedef typename GM::F FM;
  typedef typename FS::Base BS; typedef typename FM::Base BM;
  constexpr int L = FS::LANES;
  if ((int)threadIdx.x >= lanes_used || (int)threadIdx.x >= (64 / L) * L) return;
  const uint32_t item = threadIdx.x / L + 64u * blockIdx.x;
  uint32_t s = seed + 977u * threadIdx.x + 31337u * blockIdx.x;   // per LANE: coefficients differ inside an item
  uint32_t si = seed ^ (0x9E3779B9u * (item + 1u));               // per ITEM: decisions are uniform inside an item
  for (int it = 0; it < rounds; it++) {
    si = si * 1664525u + 1013904223u;
    const uint32_t choice = (si >> 8) % 8u;
    BS c[8];
    for (int k = 0; k < 8; k++) c[k] = rnd<BS>(s);
    if (choice == 7) continue;                                     // this item sits the round out (random active-item mask)
    const FS xs = FS{c[0]}, ys = FS{c[1]};
    const FM xm = FM{conv<BM>(c[0])}, ym = FM{conv<BM>(c[1])};
    Jac<FS> ps = {FS{c[2]}, FS{c[3]}, FS{c[4]}}, qs = {FS{c[5]}, FS{c[6]}, FS{c[7]}};
    Jac<FM> pm = {FM{conv<BM>(c[2])}, FM{conv<BM>(c[3])}, FM{conv<BM>(c[4])}};
    Jac<FM> qm = {FM{conv<BM>(c[5])}, FM{conv<BM>(c[6])}, FM{conv<BM>(c[7])}};
    Aff<FS> as = {xs, ys}; Aff<FM> am = {xm, ym};
    switch (choice) {
      case 0:  // field product and square
        if (!same(conv<BS>((xm * ym).c), (xs * ys).c)) atomicAdd(bad + 0, 1);
        if (!same(conv<BS>(xm.sqr().c), xs.sqr().c)) atomicAdd(bad + 1, 1);
        break;
      case 1:  // mixed addition
        if (!same_pt<FS, FM>(EC<GM>::madd(pm, am), EC<GS>::madd(ps, as))) atomicAdd(bad + 2, 1);
        break;
      case 2:  // full addition
        if (!same_pt<FS, FM>(EC<GM>::add(pm, qm), EC<GS>::add(ps, qs))) atomicAdd(bad + 3, 1);
        break;
      case 3:  // doubling, then an addition that depends on it
        if (!same_pt<FS, FM>(EC<GM>::add(EC<GM>::dbl(pm), qm), EC<GS>::add(EC<GS>::dbl(ps), qs))) atomicAdd(bad + 4, 1);
        break;
      case 4: {  // equal operands: the doubling branch INSIDE add (H = 0, r = 0)
        if (!same_pt<FS, FM>(EC<GM>::add(pm, pm), EC<GS>::add(ps, ps))) atomicAdd(bad + 5, 1);
        break;
      }
      case 5: {  // opposite operands: the cancellation branch (result = infinity: Z = 0 in both forms)
        const Jac<FM> rm = EC<GM>::add(pm, EC<GM>::neg(pm));
        const Jac<FS> rs = EC<GS>::add(ps, EC<GS>::neg(ps));
        if (rm.is_inf() != rs.is_inf() || !rs.is_inf()) atomicAdd(bad + 6, 1);
        break;
      }
      default: {  // an operand at infinity on either side, and madd of the identity
        Jac<FS> is = Jac<FS>::infinity(); Jac<FM> im = Jac<FM>::infinity();
        bool ok = same_pt<FS, FM>(EC<GM>::add(im, qm), EC<GS>::add(is, qs)) && same_pt<FS, FM>(EC<GM>::add(pm, im), EC<GS>::add(ps, is));
        const Jac<FM> mm = EC<GM>::madd(im, am); const Jac<FS> ms = EC<GS>::madd(is, as);
        ok = ok && same_pt<FS, FM>(mm, ms);
        if (!ok) atomicAdd(bad + 7, 1);
      }
    }
  }
}

// msm_merge_ones_kernel's shape: ONE item, LANES active lanes, bucket[1] += bucket[key], stored as computed
template <class GA>
__global__ void __launch_bounds__(64) merge_like(uint32_t* __restrict__ buckets, uint32_t key) {
  typedef typename GA::F F;
  if (blockIdx.x != 0 || threadIdx.x >= F::LANES) return;
  Jac<F> a = Jac<F>::load(buckets + (size_t)1 * Jac<F>::WORDS);
  Jac<F> b = Jac<F>::load(buckets + (size_t)key * Jac<F>::WORDS);
  EC<GA>::add(a, b).store(buckets + (size_t)1 * Jac<F>::WORDS);
}
// canonical images of n field elements (one lane per element), so that the host can compare values
template <class B>
__global__ void canon_kernel(uint32_t* w, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  B::load(w + (size_t)i * B::N).canonical().store(w + (size_t)i * B::N);
}

template <class GS, class GM> static int run_check(int rounds, int lanes_used, uint32_t seed, uint32_t* out8) {
  uint32_t* bad;
  if (hipMalloc(&bad, 32) != hipSuccess) return -1;
  (void)hipMemset(bad, 0, 32);
  hipLaunchKernelGGL((check<GS, GM>), dim3(8), dim3(64), 0, 0, bad, rounds, lanes_used, seed);
  const hipError_t e = hipDeviceSynchronize();
  (void)hipMemcpy(out8, bad, 32, hipMemcpyDeviceToHost);
  (void)hipFree(bad);
  return e == hipSuccess ? 0 : -2;
}
// returns the number of differing canonical words of bucket[1] between the two forms (0 = equal), < 0 on a HIP error
template <class GS, class GM> static int run_merge(uint32_t seed) {
  typedef typename GS::F FS;
  constexpr int PW = Jac<FS>::WORDS, N = FS::Base::N;
  std::vector<uint32_t> h(8 * PW);
  uint32_t s = seed;
  for (auto& w : h) { s = s * 1664525u + 1013904223u; w = (s >> 4) & 0x0FFFFFFFu; }
  for (int i = 0; i < 8 * PW; i += N) h[i + N - 1] &= 0xFFFFu;
  uint32_t *d1, *d2;
  if (hipMalloc(&d1, h.size() * 4) != hipSuccess || hipMalloc(&d2, h.size() * 4) != hipSuccess) return -1;
  (void)hipMemcpy(d1, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d2, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((merge_like<GS>), dim3(1), dim3(64), 0, 0, d1, 5u);
  hipLaunchKernelGGL((merge_like<GM>), dim3(1), dim3(64), 0, 0, d2, 5u);
  const uint32_t nel = 8 * PW / N;
  hipLaunchKernelGGL((canon_kernel<typename FS::Base>), dim3((nel + 63) / 64), dim3(64), 0, 0, d1, nel);
  hipLaunchKernelGGL((canon_kernel<typename FS::Base>), dim3((nel + 63) / 64), dim3(64), 0, 0, d2, nel);
  std::vector<uint32_t> r1(h.size()), r2(h.size());
  const hipError_t e = hipDeviceSynchronize();
  (void)hipMemcpy(r1.data(), d1, h.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(r2.data(), d2, h.size() * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d1); (void)hipFree(d2);
  if (e != hipSuccess) return -2;
  int diff = 0;
  for (size_t i = 0; i < h.size(); i++) diff += r1[i] != r2[i];
  return diff;
}

typedef G2Cfg3S<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3, false> Q3S;
typedef G2Cfg3SMB<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3> Q3M;
typedef G2Cfg2S<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2, false> Q2S;
typedef G2Cfg2SMB<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2> Q2M;

// which: 3 = Fq3-753 (MNT6-753 G2), 2 = Fq2-753 (MNT4-753 G2).  out8 = mismatch counts of the eight checks.
extern "C" int gc_mailbox_check(int which, int rounds, int lanes_used, uint32_t seed, uint32_t* out8) {
  if (which == 3) return run_check<Q3S, Q3M>(rounds, lanes_used, seed, out8);
  if (which == 2) return run_check<Q2S, Q2M>(rounds, lanes_used, seed, out8);
  return -3;
}
extern "C" int gc_mailbox_merge(int which, uint32_t seed) {
  if (which == 3) return run_merge<Q3S, Q3M>(seed);
  if (which == 2) return run_merge<Q2S, Q2M>(seed);
  return -3;
}
