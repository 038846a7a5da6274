// Developer check (GPU): the LDS-mailbox variants of the lane-split extension fields against the plain ones, operation by operation,
// on pseudo-random operands: field product / square, and the group operations the MSM kernels use (madd, add, dbl).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I pcd_amd/csrc tools/microbench/k2_mailbox_check.hip -o build/k2_mailbox_check && build/k2_mailbox_check
#include <cstdio>
#include <vector>
#include "common.h"
using namespace pcd;

template <class FD, class FS> __device__ FD conv(const FS& a) { FD r; for (int i = 0; i < FS::N; i++) r.v[i] = a.v[i]; return r; }
template <class B> __device__ B rnd(uint32_t& s) {
  B r;
  for (int i = 0; i < B::N; i++) { s = s * 1664525u + 1013904223u; r.v[i] = (s >> 4) & 0x0FFFFFFFu; }
  r.v[B::N - 1] &= 0xFFFFu;  // < 2^(28 (N-1) + 16) < p
  return r;
}
template <class B> __device__ bool same(const B& a0, const B& b0) { const B a = a0.canonical(), b = b0.canonical(); bool ok = true; for (int i = 0; i < B::N; i++) ok &= a.v[i] == b.v[i]; return ok; }  // equal mod p (representatives in [0, 2p) may differ)

// GS: plain split config, GM: mailbox split config (same lane layout)
template <class GS, class GM>
__global__ void __launch_bounds__(64) check(uint32_t* bad, int rounds, int lanes_used) {
  typedef typename GS::F FS; typedef typename GM::F FM;
  typedef typename FS::Base BS; typedef typename FM::Base BM;
  if ((int)threadIdx.x >= lanes_used) return;
  uint32_t s = 12345u + 977u * threadIdx.x + 31337u * blockIdx.x;
  for (int it = 0; it < rounds; it++) {
    BS c[8];
    for (int k = 0; k < 8; k++) c[k] = rnd<BS>(s);
    const FS xs = FS{c[0]}, ys = FS{c[1]};
    const FM xm = FM{conv<BM>(c[0])}, ym = FM{conv<BM>(c[1])};
    if (!same(conv<BS>((xm * ym).c), (xs * ys).c)) atomicAdd(bad + 0, 1);
    if (!same(conv<BS>(xm.sqr().c), xs.sqr().c)) atomicAdd(bad + 1, 1);
    Jac<FS> ps = {FS{c[2]}, FS{c[3]}, FS{c[4]}}, qs = {FS{c[5]}, FS{c[6]}, FS{c[7]}};
    Jac<FM> pm = {FM{conv<BM>(c[2])}, FM{conv<BM>(c[3])}, FM{conv<BM>(c[4])}};
    Jac<FM> qm = {FM{conv<BM>(c[5])}, FM{conv<BM>(c[6])}, FM{conv<BM>(c[7])}};
    Aff<FS> as = {xs, ys}; Aff<FM> am = {xm, ym};
    { auto a = EC<GS>::madd(ps, as); auto b = EC<GM>::madd(pm, am);
      if (!(same(conv<BS>(b.X.c), a.X.c) && same(conv<BS>(b.Y.c), a.Y.c) && same(conv<BS>(b.Z.c), a.Z.c))) atomicAdd(bad + 2, 1); }
    { auto a = EC<GS>::add(ps, qs); auto b = EC<GM>::add(pm, qm);
      if (!(same(conv<BS>(b.X.c), a.X.c) && same(conv<BS>(b.Y.c), a.Y.c) && same(conv<BS>(b.Z.c), a.Z.c))) atomicAdd(bad + 3, 1); }
    { auto a = EC<GS>::dbl(ps); auto b = EC<GM>::dbl(pm);
      if (!(same(conv<BS>(b.X.c), a.X.c) && same(conv<BS>(b.Y.c), a.Y.c) && same(conv<BS>(b.Z.c), a.Z.c))) atomicAdd(bad + 4, 1); }
    // divergent use: only some items take a branch with products in it (the reduction levels do this)
    if (((threadIdx.x / FS::LANES) + it) % 3 == 0) {
      auto a = EC<GS>::add(EC<GS>::dbl(ps), qs); auto b = EC<GM>::add(EC<GM>::dbl(pm), qm);
      if (!(same(conv<BS>(b.X.c), a.X.c) && same(conv<BS>(b.Y.c), a.Y.c) && same(conv<BS>(b.Z.c), a.Z.c))) atomicAdd(bad + 5, 1);
    }
  }
}

// msm_merge_ones_kernel verbatim, on a given configuration: bucket[1] += bucket[key]
template <class GA, bool CANON>
__global__ void __launch_bounds__(64) merge_like(uint32_t* __restrict__ buckets, uint32_t key) {
  typedef typename GA::F F;
  if (blockIdx.x != 0 || threadIdx.x >= F::LANES) return;
  Jac<F> a = Jac<F>::load(buckets + (size_t)1 * Jac<F>::WORDS);
  Jac<F> b = Jac<F>::load(buckets + (size_t)key * Jac<F>::WORDS);
  Jac<F> r = EC<GA>::add(a, b);
  if (CANON) { r.X = F{r.X.c.canonical()}; r.Y = F{r.Y.c.canonical()}; r.Z = F{r.Z.c.canonical()}; }  // (values, not representatives)
  r.store(buckets + (size_t)1 * Jac<F>::WORDS);
}
template <class GS, class GM, bool CANON> static void run_merge(const char* name) {
  typedef typename GS::F FS;
  constexpr int PW = Jac<FS>::WORDS;
  std::vector<uint32_t> h(8 * PW);
  uint32_t s = 4242u;
  for (auto& w : h) { s = s * 1664525u + 1013904223u; w = (s >> 4) & 0x0FFFFFFFu; }
  for (int i = 0; i < 8 * PW; i += FS::Base::N) h[i + FS::Base::N - 1] &= 0xFFFFu;
  uint32_t *d1, *d2; (void)hipMalloc(&d1, h.size() * 4); (void)hipMalloc(&d2, h.size() * 4);
  (void)hipMemcpy(d1, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d2, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((merge_like<GS, CANON>), dim3(1), dim3(64), 0, 0, d1, 5u);
  hipLaunchKernelGGL((merge_like<GM, CANON>), dim3(1), dim3(64), 0, 0, d2, 5u);
  std::vector<uint32_t> r1(h.size()), r2(h.size());
  (void)hipMemcpy(r1.data(), d1, h.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(r2.data(), d2, h.size() * 4, hipMemcpyDeviceToHost);
  int diff = 0; for (size_t i = 0; i < h.size(); i++) diff += r1[i] != r2[i];
  printf("%s merge-like kernel: %d differing words of %d (plain vs mailbox)\n", name, diff, PW);
  for (int c = 0; c < 3 * FS::DEG; c++) { int dd = 0; for (int i = 0; i < FS::Base::N; i++) dd += r1[PW + c * FS::Base::N + i] != r2[PW + c * FS::Base::N + i]; printf("  element %d: %d", c, dd);
    if (dd && dd <= 3) for (int i = 0; i < FS::Base::N; i++) if (r1[PW + c * FS::Base::N + i] != r2[PW + c * FS::Base::N + i]) printf("   limb %d: %08x vs %08x", i, r1[PW + c * FS::Base::N + i], r2[PW + c * FS::Base::N + i]);
    printf("\n"); }
  // the limbs of the first Y coefficient in both forms (to be checked against each other mod p off line)
  printf("  Y.c0 plain  :"); for (int i = 0; i < FS::Base::N; i++) printf(" %08x", r1[PW + 3 * FS::Base::N + i]); printf("\n");
  printf("  Y.c0 mailbox:"); for (int i = 0; i < FS::Base::N; i++) printf(" %08x", r2[PW + 3 * FS::Base::N + i]); printf("\n");
  (void)hipFree(d1); (void)hipFree(d2);
}
template <class GS, class GM> static void run(const char* name, int lanes_used) {
  uint32_t* bad; (void)hipMalloc(&bad, 32); (void)hipMemset(bad, 0, 32);
  hipLaunchKernelGGL((check<GS, GM>), dim3(8), dim3(64), 0, 0, bad, 8, lanes_used);
  uint32_t h[6]; (void)hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost);
  printf("%s: mismatches mul=%u sqr=%u madd=%u add=%u dbl=%u divergent=%u  (%s)\n", name, h[0], h[1], h[2], h[3], h[4], h[5], hipGetErrorString(hipGetLastError()));
  (void)hipFree(bad);
}
int main() {
  run<G2Cfg3S<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3, false>, G2Cfg3SMB<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3>>("Fq3-753 split", 63);
  run<G2Cfg3S<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3, false>, G2Cfg3SMB<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3>>("Fq3-753 split, one item", 3);
  run<G2Cfg3S<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3, false>, G2Cfg3SMB<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3>>("Fq3-753 split, ten items", 30);
  run<G2Cfg2S<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2, false>, G2Cfg2SMB<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2>>("Fq2-753 split", 64);
  run_merge<G2Cfg3S<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3, false>, G2Cfg3SMB<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3>, true>("Fq3-753 canonical");
  run_merge<G2Cfg3S<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3, false>, G2Cfg3SMB<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3>, false>("Fq3-753 as stored");
  run_merge<G2Cfg2S<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2, false>, G2Cfg2SMB<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2>, false>("Fq2-753 as stored");
  return 0;
}
