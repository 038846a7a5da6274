// TEST HARNESS ONLY (never part of libpcdhip.so): compiles the __host__ __device__ field / curve
// templates of pcd_amd/csrc for the HOST so that their formulas can be checked against the oracle in
// this GPU-less container.  It exercises no kernel and is not a CPU fallback of anything.
#include "../../pcd_amd/csrc/ec.hip.h"
using namespace pcd;

template <class G>
static void msm_naive(const uint32_t* bases, const uint32_t* scalars, int n, uint32_t* out) {
  typedef typename G::F F;
  typedef EC<G> E;
  Jac<F> acc = Jac<F>::infinity();
  constexpr int NS = G::FR::N32;
  for (int i = 0; i < n; i++) {
    Aff<F> p = Aff<F>::from_abi(bases + (size_t)i * Aff<F>::ABI_WORDS);
    if (p.is_inf()) continue;
    Jac<F> q = E::mul(Jac<F>{p.x, p.y, F::one()}, scalars + (size_t)i * NS, NS);
    acc = E::add(acc, q);
    // also exercise madd with the running sum
    acc = E::madd(acc, p);
    acc = E::add(acc, E::neg(Jac<F>{p.x, p.y, F::one()}));
  }
  acc.to_abi(out);
}

// field-level checks: out[0..] = a*b, a+b, a-b, inv(a), a*17 (mul_small), canonical words of a, ..., inv_gcd(a) twice  (ABI images)
template <class P>
static void field_ops(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  typedef Fp<P> F;
  F x = F::from_abi(a), y = F::from_abi(b);
  constexpr int W = F::ABI_WORDS;
  (x * y).to_abi(out);
  (x + y).to_abi(out + W);
  (x - y).to_abi(out + 2 * W);
  x.inv_fermat().to_abi(out + 3 * W);
  x.mul_small(17).to_abi(out + 4 * W);
  x.to_canonical_words(out + 5 * W);
  ((x + y + y - x - x).dbl().neg().mul_small(121) * F::from_canonical_words(out + 5 * W)).to_abi(out + 6 * W);
  x.inv_gcd().to_abi(out + 7 * W);            // the divstep inverse
  (x + F::one() - F::one()).inv_gcd().to_abi(out + 8 * W);  // ... of another representative of the same value
}
extern "C" int hc_field_ops(int field, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  switch (field) {
    case 0: field_ops<F298A>(a, b, out); break;
    case 1: field_ops<F298B>(a, b, out); break;
    case 2: field_ops<F753A>(a, b, out); break;
    case 3: field_ops<F753B>(a, b, out); break;
    default: return -1;
  }
  return 0;
}

// raw 28-bit-limb images (N words each, values in [0, 2p)): out = (a + b) || (a - b) for n pairs -- the operands are NOT converted,
// so the cases the two-limb estimate of Fp::operator+ / operator- cannot decide can be hit on purpose
template <class P_>
static void addsub_raw(const uint32_t* a, const uint32_t* b, int n, uint32_t* out) {
  typedef Fp<P_> F;
  for (int k = 0; k < n; k++) {
    const F x = F::load(a + (size_t)k * F::N), y = F::load(b + (size_t)k * F::N);
    (x + y).store(out + (size_t)(2 * k) * F::N);
    (x - y).store(out + (size_t)(2 * k + 1) * F::N);
  }
}
extern "C" int hc_addsub_raw(int field, const uint32_t* a, const uint32_t* b, int n, uint32_t* out) {
  switch (field) {
    case 0: addsub_raw<F298A>(a, b, n, out); break;
    case 1: addsub_raw<F298B>(a, b, n, out); break;
    case 2: addsub_raw<F753A>(a, b, n, out); break;
    case 3: addsub_raw<F753B>(a, b, n, out); break;
    default: return -1;
  }
  return 0;
}

extern "C" int hc_msm_naive(int group_idx, const uint32_t* bases, const uint32_t* scalars, int n, uint32_t* out) {
  switch (group_idx) {
    case 0: msm_naive<G1_MNT4_298>(bases, scalars, n, out); break;
    case 1: msm_naive<G2_MNT4_298>(bases, scalars, n, out); break;
    case 2: msm_naive<G1_MNT6_298>(bases, scalars, n, out); break;
    case 3: msm_naive<G2_MNT6_298>(bases, scalars, n, out); break;
    case 4: msm_naive<G1_MNT4_753>(bases, scalars, n, out); break;
    case 5: msm_naive<G2_MNT4_753>(bases, scalars, n, out); break;
    case 6: msm_naive<G1_MNT6_753>(bases, scalars, n, out); break;
    case 7: msm_naive<G2_MNT6_753>(bases, scalars, n, out); break;
    default: return -1;
  }
  return 0;
}

// chain of mixed additions acc += P_i with the lazily reduced accumulator (EC::madd_lz) and with the ordinary madd:
// out = lazy result || ordinary result (Jacobian, C-ABI image); points may repeat (doubling branch) and cancel
template <class G>
static void madd_chain(const uint32_t* bases, int n, uint32_t* out) {
  typedef typename G::F F;
  typedef EC<G> E;
  typename E::AccLz lz = E::lz_infinity();
  Jac<F> acc = Jac<F>::infinity();
  for (int i = 0; i < n; i++) {
    Aff<F> p = Aff<F>::from_abi(bases + (size_t)i * Aff<F>::ABI_WORDS);
    lz = E::madd_lz(lz, p);
    acc = E::madd(acc, p);
  }
  E::lz_to_jac(lz).to_abi(out);
  acc.to_abi(out + Jac<F>::ABI_WORDS);
}
extern "C" int hc_madd_chain(int curve, const uint32_t* bases, int n, uint32_t* out) {
  switch (curve) {
    case 0: madd_chain<G1_MNT4_298>(bases, n, out); break;
    case 1: madd_chain<G1_MNT6_298>(bases, n, out); break;
    default: return -1;
  }
  return 0;
}

// the same chain in plain XYZZ coordinates (EC::madd_x: what the accumulation of every other group runs on) against the Jacobian madd
template <class G>
static void maddx_chain(const uint32_t* bases, int n, uint32_t* out) {
  typedef typename G::F F;
  typedef EC<G> E;
  typename E::AccX ax = E::x_infinity();
  Jac<F> acc = Jac<F>::infinity();
  for (int i = 0; i < n; i++) {
    Aff<F> p = Aff<F>::from_abi(bases + (size_t)i * Aff<F>::ABI_WORDS);
    ax = E::madd_x(ax, p);
    acc = E::madd(acc, p);
  }
  E::x_to_jac(ax).to_abi(out);
  acc.to_abi(out + Jac<F>::ABI_WORDS);
}
extern "C" int hc_maddx_chain(int group_idx, const uint32_t* bases, int n, uint32_t* out) {
  switch (group_idx) {
    case 0: maddx_chain<G1_MNT4_298>(bases, n, out); break;
    case 1: maddx_chain<G2_MNT4_298>(bases, n, out); break;
    case 2: maddx_chain<G1_MNT6_298>(bases, n, out); break;
    case 3: maddx_chain<G2_MNT6_298>(bases, n, out); break;
    case 4: maddx_chain<G1_MNT4_753>(bases, n, out); break;
    case 5: maddx_chain<G2_MNT4_753>(bases, n, out); break;
    case 6: maddx_chain<G1_MNT6_753>(bases, n, out); break;
    case 7: maddx_chain<G2_MNT6_753>(bases, n, out); break;
    default: return -1;
  }
  return 0;
}

