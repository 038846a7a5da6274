// TEST HARNESS ONLY (never part of libpcdhip.so): compiles the __host__ __device__ field / curve
// templates of pcd_amd/csrc for the HOST so that their formulas can be checked against the oracle in
// this GPU-less container.  It exercises no kernel and is not a CPU fallback of anything.
#include "../../pcd_amd/csrc/ec.cuh"
using namespace pcd;

template <class G>
static void msm_naive(const uint32_t* bases, const uint32_t* scalars, int n, uint32_t* out) {
  typedef typename G::F F;
  typedef EC<G> E;
  Jac<F> acc = Jac<F>::infinity();
  constexpr int NS = G::FR::N;
  for (int i = 0; i < n; i++) {
    Aff<F> p = Aff<F>::load(bases + (size_t)i * Aff<F>::WORDS);
    if (p.is_inf()) continue;
    Jac<F> q = E::mul(Jac<F>{p.x, p.y, F::one()}, scalars + (size_t)i * NS, NS);
    acc = E::add(acc, q);
    // also exercise madd with the running sum
    acc = E::madd(acc, p);
    acc = E::add(acc, E::neg(Jac<F>{p.x, p.y, F::one()}));
  }
  acc.store(out);
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

#include "../../pcd_amd/csrc/pairing.cuh"
template <class PC>
static void pairing_host(const uint32_t* g1, const uint32_t* g2, uint32_t* out) {
  typedef Pairing<PC> PE;
  typename PE::Frob t;
  frob_init<typename PE::Fq, PE::K, PC::NR>(t);
  auto f = PE::miller_loop(Aff<typename PE::Fq>::load(g1), Aff<typename PE::E>::load(g2));
  PE::final_exponentiation(f, t).store(out);
}
extern "C" int hc_pairing(int curve, const uint32_t* g1, const uint32_t* g2, uint32_t* out) {
  switch (curve) {
    case 0: pairing_host<PC_MNT4_298>(g1, g2, out); break;
    case 1: pairing_host<PC_MNT6_298>(g1, g2, out); break;
    case 2: pairing_host<PC_MNT4_753>(g1, g2, out); break;
    case 3: pairing_host<PC_MNT6_753>(g1, g2, out); break;
    default: return -1;
  }
  return 0;
}
