// TEST HARNESS ONLY (never part of libpcdhip.so): the multi-lane group operations of the bucket reduction (round 6) against the plain
// ones, operation by operation, on the GPU:
//   EC4::add4 / dbl4   four lanes per operation, prime-field groups (the late pair levels)
//   EC2::add2 / dbl2   two lanes per operation (prime-field groups) and two HALVES of L lanes per operation (the lane-split Fq2 / Fq3 groups,
//                      inlined 298-bit and mailbox 753-bit forms)
// under RANDOM ACTIVE-ITEM MASKS and the DIVERGENT branches an MSM rarely meets in its tail: equal operands (the doubling branch inside an
// addition), opposite operands (cancellation to infinity), an operand at infinity on either side, a doubling of infinity.  Both forms run
// the same formulas, so the coordinates must agree as VALUES mod p (representatives in [0, 2p) may differ), in every lane of an item.
// Built by __graft_entry__.build() (hipcc, gfx950) into tests/gpucheck/libgpucheck_ml.so; tests/test_gpu_multilane.py drives it.
#include <cstdio>
#include "../../pcd_amd/csrc/common.h"
using namespace pcd;

template <class B> __device__ B ml_rnd(uint32_t& s) {
  B r;
  for (int i = 0; i < B::N; i++) { s = s * 1664525u + 1013904223u; r.v[i] = (s >> 4) & 0x0FFFFFFFu; }
  r.v[B::N - 1] &= 0xFFFFu;  // < 2^(28 (N-1) + 16) < p
  return r;
}
template <class B> __device__ bool ml_same(const B& a0, const B& b0) {
  const B a = a0.canonical(), b = b0.canonical();
  bool ok = true;
  for (int i = 0; i < B::N; i++) ok &= a.v[i] == b.v[i];
  return ok;
}
template <class F> struct MlBase { typedef F type; __device__ static F mk(const F& c) { return c; } __device__ static const F& of(const F& x) { return x; } };
template <class B, unsigned NR> struct MlBase<Fp2S<B, NR>> { typedef B type; __device__ static Fp2S<B, NR> mk(const B& c) { return {c}; } __device__ static const B& of(const Fp2S<B, NR>& x) { return x.c; } };
template <class B, unsigned NR> struct MlBase<Fp3S<B, NR>> { typedef B type; __device__ static Fp3S<B, NR> mk(const B& c) { return {c}; } __device__ static const B& of(const Fp3S<B, NR>& x) { return x.c; } };
template <class F> __device__ bool ml_same_pt(const Jac<F>& a, const Jac<F>& b) {
  typedef MlBase<F> M;
  // (an identity is Z = 0: X and Y are then never looked at)
  if (a.is_inf() || b.is_inf()) return a.is_inf() == b.is_inf();
  return ml_same(M::of(a.X), M::of(b.X)) && ml_same(M::of(a.Y), M::of(b.Y)) && ml_same(M::of(a.Z), M::of(b.Z));
}
template <class G, int MODE> struct MlOps;   // MODE 4: EC4, 2: EC2
template <class G> struct MlOps<G, 4> { typedef Jac<typename G::F> J; __device__ static J add(const J& a, const J& b) { return EC4<G>::add4(a, b); } __device__ static J dbl(const J& a) { return EC4<G>::dbl4(a); } };
template <class G> struct MlOps<G, 2> { typedef Jac<typename G::F> J; __device__ static J add(const J& a, const J& b) { return EC2<G>::add2(a, b); } __device__ static J dbl(const J& a) { return EC2<G>::dbl2(a); } };

template <class G, int MODE>
__global__ void __launch_bounds__(64) ml_check(uint32_t* bad, int rounds, uint32_t seed) {
  typedef typename G::F F;
  typedef MlBase<F> M;
  typedef typename M::type B;
  typedef MlOps<G, MODE> O;
  typedef EC<G> E;
  constexpr int L = HalfLanes<F>::L;                  // lanes per point (1: prime field)
  constexpr int IL = MODE == 4 ? 4 : 2 * L;           // lanes per item
  constexpr int PER = 64 / IL;
  if ((int)threadIdx.x >= PER * IL) return;
  const uint32_t item = threadIdx.x / IL + (uint32_t)PER * blockIdx.x;
  const uint32_t role = (threadIdx.x & 63u) % (uint32_t)L;    // which coefficient of a lane-split value this lane holds
  uint32_t s = seed ^ (0x9E3779B9u * (item * 8u + role + 1u));  // equal in the lanes of an item that hold the same coefficient
  uint32_t si = seed ^ (0x85EBCA6Bu * (item + 1u));           // decisions: uniform inside an item
  for (int it = 0; it < rounds; it++) {
    si = si * 1664525u + 1013904223u;
    const uint32_t choice = (si >> 8) % 8u;
    B c[6];
    for (int k = 0; k < 6; k++) c[k] = ml_rnd<B>(s);
    if (choice == 7) continue;                                 // this item sits the round out
    const Jac<F> p = {M::mk(c[0]), M::mk(c[1]), M::mk(c[2])}, q = {M::mk(c[3]), M::mk(c[4]), M::mk(c[5])};
    const Jac<F> inf = Jac<F>::infinity();
    switch (choice) {
      case 0: case 1:
        if (!ml_same_pt(O::add(p, q), E::add(p, q))) atomicAdd(bad + 0, 1);
        break;
      case 2:
        if (!ml_same_pt(O::dbl(p), E::dbl(p))) atomicAdd(bad + 1, 1);
        break;
      case 3:  // a doubling, then an addition that depends on it, then a doubling of that
        if (!ml_same_pt(O::dbl(O::add(O::dbl(p), q)), E::dbl(E::add(E::dbl(p), q)))) atomicAdd(bad + 2, 1);
        break;
      case 4:  // equal operands: the doubling branch INSIDE the addition
        if (!ml_same_pt(O::add(p, p), E::add(p, p))) atomicAdd(bad + 3, 1);
        break;
      case 5: {  // opposite operands: cancellation
        const Jac<F> r = O::add(p, E::neg(p));
        if (!r.is_inf()) atomicAdd(bad + 4, 1);
        break;
      }
      default: {  // identities on either side, doubling of the identity
        bool ok = ml_same_pt(O::add(inf, q), q) && ml_same_pt(O::add(p, inf), p) && O::dbl(inf).is_inf() && O::add(inf, inf).is_inf();
        if (!ok) atomicAdd(bad + 5, 1);
      }
    }
  }
  atomicAdd(bad + 7, 1);   // lanes that ran to the end (the host checks that the kernel did run)
}

template <class G, int MODE> static int ml_run(int rounds, uint32_t seed, uint32_t* out8) {
  uint32_t* bad;
  if (hipMalloc(&bad, 32) != hipSuccess) return -1;
  (void)hipMemset(bad, 0, 32);
  hipLaunchKernelGGL((ml_check<G, MODE>), dim3(8), dim3(64), 0, 0, bad, rounds, seed);
  const hipError_t e = hipDeviceSynchronize();
  (void)hipMemcpy(out8, bad, 32, hipMemcpyDeviceToHost);
  (void)hipFree(bad);
  return e == hipSuccess ? 0 : -2;
}

typedef G1CfgMB<F753A, F753B, PCD_MNT4_753_A_SMALL, 2> G1M753;
typedef G2Cfg2S<F298A, F298B, PCD_MNT4_298_A_SMALL, PCD_MNT4_298_NR_SMALL, 0, true> Q2S298;
typedef G2Cfg3S<F298B, F298A, PCD_MNT6_298_A_SMALL, PCD_MNT6_298_NR_SMALL, 1, true> Q3S298;
typedef G2Cfg2SMB<F753A, F753B, PCD_MNT4_753_A_SMALL, PCD_MNT4_753_NR_SMALL, 2> Q2M753;
typedef G2Cfg3SMB<F753B, F753A, PCD_MNT6_753_A_SMALL, PCD_MNT6_753_NR_SMALL, 3> Q3M753;

// which: 0 G1-298 EC4, 1 G1-298 EC2, 2 G1-753 (mailbox) EC4, 3 G1-753 (mailbox) EC2, 4 Fq2-298 halves, 5 Fq3-298 halves, 6 Fq2-753 (mailbox) halves,
// 7 Fq3-753 (mailbox) halves.  out8[0..5] = mismatch counts of the six checks, out8[7] = lanes that finished.
extern "C" int gc_multilane_check(int which, int rounds, uint32_t seed, uint32_t* out8) {
  switch (which) {
    case 0: return ml_run<G1_MNT4_298, 4>(rounds, seed, out8);
    case 1: return ml_run<G1_MNT4_298, 2>(rounds, seed, out8);
    case 2: return ml_run<G1M753, 4>(rounds, seed, out8);
    case 3: return ml_run<G1M753, 2>(rounds, seed, out8);
    case 4: return ml_run<Q2S298, 2>(rounds, seed, out8);
    case 5: return ml_run<Q3S298, 2>(rounds, seed, out8);
    case 6: return ml_run<Q2M753, 2>(rounds, seed, out8);
    case 7: return ml_run<Q3M753, 2>(rounds, seed, out8);
  }
  return -3;
}
